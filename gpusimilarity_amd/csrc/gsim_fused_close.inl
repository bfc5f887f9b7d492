// gsim_fused_close.inl -- phase 5 of the single launch, a piece of fused_kernel's body (included there): edge E4 (closing tickets, the
// result header as completion signal, the reset of the per-query state) of gsim_fused_protocol.h.
    // ---- 5. the last selector closes the query -----------------------------------------------
    GSIM_STAMP(6);
    if (fa.done_flag) { // (wave-uniform; most threads wrote nothing)
        const uint32_t wsum = wave_sum_dpp(cks);
        if (lane == 0 && wsum) atomicAdd(&sh.cks, wsum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave: its hits have left the CU
    __syncthreads();
    if (tid == 0) {
        // The hits were stored write-through at system scope (sc0 sc1) and every wave has waited for their
        // acknowledgements: they are in memory, there is nothing for a release fence to write back.  (Plain stores need
        // the fence -- 16 of 600 k queries came back incomplete without it, DESIGN.md 7 (g) -- and it cost 1.3 us per
        // query.  GSIM_FUSED_FLAGS=1024 puts it back.)
        if (fa.xflags & 1024u) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the ticket also carries "this selector saw the query fail" (bit 16 up): the closer learns it without another
        // round trip (a workgroup that set QueryState::redo while publishing did so before the grid-wide wait: every
        // selector read it after the wait and is not `good`)
        // Two levels, as the arrival: 256 atomics on ONE word queue up behind each other (the last ticket came 3.3 us
        // after the average selector was ready); a ticket per group b % 8 first, the group's last adds to the top.
        const uint32_t x = blockIdx.x % 8u;
        const uint32_t group_size = (nwg - x + 7u) / 8u, ngroups = nwg < 8u ? nwg : 8u;
        // (64-bit tickets: count in bits 0..15, failures in 16..31, the checksum of the hits written so far in 32..63 -- one
        // atomic carries all three, so the last holder knows the sum without another round trip)
        const u64 mine64 = (static_cast<u64>(sh.cks) << 32) | (good ? 1ull : 0x10001ull);
        const u64 tg = atomicAdd(reinterpret_cast<u64*>(&fa.arrive[(17u + x) * 32u]), mine64);
        uint32_t closing = 0, failed = 0;
        if ((static_cast<uint32_t>(tg) & 0xFFFFu) == group_size - 1u) {
            const bool gfail = ((static_cast<uint32_t>(tg) >> 16) & 0xFFFFu) != 0 || !good;
            const uint32_t gcks = static_cast<uint32_t>((tg + mine64) >> 32);
            const u64 top64 = (static_cast<u64>(gcks) << 32) | (gfail ? 0x10001ull : 1ull);
            const u64 tt = atomicAdd(&st->sel_done, top64);
            closing = (static_cast<uint32_t>(tt) & 0xFFFFu) == ngroups - 1u ? 1u : 0u;
            failed = (((static_cast<uint32_t>(tt) >> 16) & 0xFFFFu) != 0 || gfail) ? 1u : 0u;
            sh.cks_total = static_cast<uint32_t>((tt + top64) >> 32);
        }
        sh.ticket = closing | (failed << 1);
    }
    __syncthreads();
    GSIM_STAMP(7);
    if (!(sh.ticket & 1u)) return;
    const uint32_t redo = ((sh.ticket & 2u) != 0 || !good) ? 1u : 0u;
    if (redo && tid == 0) atomicOr(&st->redo, kRedoSeen); // (the gated classic kernels behind an enqueue-only launch read it)
    if (tid == 0) {
        { // the header, write-through as the hits
            const u64 approx = a.cutoff > 0.0f ? __hip_atomic_load(&st->kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.nrows;
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(fa.result, 0, 16, 0x00020000);
            // The block is complete (every selector waited for its hits before its ticket): for a synchronous caller the
            // header carries the query's epoch -- the host polls it, one 16-byte write tells it everything -- and the
            // tidying up below happens behind the caller's back.
            const uint32_t flags = (redo ? 2u : 0u) | (fa.done_flag ? fa.epoch << 8 : 0u);
            // (synchronous callers: the upper half of approx carries the block's checksum, see kBlockCheckMul)
            const uint32_t w3 = fa.done_flag ? sh.cks_total + fa.epoch * kBlockCheckMul : static_cast<uint32_t>(approx >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{redo ? 0u : (nfin < a.k ? nfin : a.k), flags, static_cast<uint32_t>(approx), w3},
                                                   rrs, 0, 0, /*sc0 sc1*/ 17);
        }
        // re-zero the per-query state for the next launch (stream-ordered behind this one)
        st->ncand_sum += __hip_atomic_load(&st->ncand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st->nfinal_sum += redo ? 0u : nfin;
        st->queries += redo ? 0u : 1u;
        st->redo_sum += redo ? 1u : 0u;
        if (redo) st->redo_why |= __hip_atomic_load(&st->redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->kept, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->ncand, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->gtau, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->elected, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->sel_done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // synchronous callers read the hand-back from the header (no gated kernels behind this launch): the next
        // launch, possibly already enqueued, starts clean
        if (fa.done_flag) __hip_atomic_store(&st->redo, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    rezero_exchange();
    if (dbg && tid == 0) fa.dbg[static_cast<u64>(gridDim.x) * 24] = wall_clock64(); // the very end
