// fsim_reader.cpp -- see fsim_reader.h.
#include "fsim_reader.h"

#include <zlib.h>

#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <thread>

#include "qds.h"

namespace gpusim
{

std::vector<unsigned char> q_uncompress(const std::vector<unsigned char>& blob)
{
    if (blob.size() < 4) return {};
    const uint32_t expect = (uint32_t(blob[0]) << 24) | (uint32_t(blob[1]) << 16) | (uint32_t(blob[2]) << 8) |
                            uint32_t(blob[3]);
    std::vector<unsigned char> out(expect ? expect : 1);
    uLongf len = static_cast<uLongf>(out.size());
    const int rc = uncompress(out.data(), &len, blob.data() + 4, static_cast<uLong>(blob.size() - 4));
    if (rc != Z_OK || len != expect) throw std::runtime_error("qUncompress: corrupt block in database file");
    out.resize(expect);
    return out;
}

namespace
{
std::vector<std::vector<unsigned char>> read_block_list(QdsReader& r)
{
    const int n = r.i32();
    if (n < 0) throw std::runtime_error("database file: negative block count");
    std::vector<std::vector<unsigned char>> blocks(static_cast<size_t>(n));
    for (auto& b : blocks) b = r.bytearray();
    return blocks;
}

void strings_from_block(const std::vector<unsigned char>& compressed, std::vector<char*>& out)
{
    const std::vector<unsigned char> raw = q_uncompress(compressed);
    QdsReader r(raw);
    while (!r.atEnd()) out.push_back(r.cstr_new());
}
} // namespace

void extractData(const std::string& database_fname, int& fp_bitcount, int& fp_count, std::string& dbkey,
                 std::vector<std::vector<char>>& fingerprint_data, std::vector<char*>& smiles_vector,
                 std::vector<char*>& ids_vector)
{
    std::ifstream f(database_fname, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open database file " + database_fname);
    std::vector<unsigned char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    QdsReader r(raw);
    const int version = r.i32();
    if (version != DATABASE_VERSION) {
        throw std::runtime_error("Database version incompatible with this GPUSim version");
    }
    dbkey = r.cstr();
    fp_bitcount = r.i32();
    fp_count = r.i32();

    auto fp_blocks = read_block_list(r);
    auto smi_blocks = read_block_list(r);
    auto id_blocks = read_block_list(r);

    fingerprint_data.assign(fp_blocks.size(), {});
    std::vector<std::vector<char*>> smiles_data(smi_blocks.size()), ids_data(id_blocks.size());
    std::vector<std::thread> pool;
    std::vector<std::string> errors(fp_blocks.size() + smi_blocks.size() + id_blocks.size());
    size_t job = 0;
    for (size_t i = 0; i < fp_blocks.size(); i++, job++) {
        std::fprintf(stderr, "  loading FP %zu of %zu\n", i + 1, fp_blocks.size());
        pool.emplace_back([&, i, job] {
            try {
                const auto u = q_uncompress(fp_blocks[i]);
                fp_blocks[i].clear();
                fingerprint_data[i].assign(u.begin(), u.end());
            } catch (const std::exception& e) {
                errors[job] = e.what();
            }
        });
    }
    for (size_t i = 0; i < smi_blocks.size(); i++, job++) {
        std::fprintf(stderr, "  loading SMI %zu of %zu\n", i + 1, smi_blocks.size());
        pool.emplace_back([&, i, job] {
            try {
                strings_from_block(smi_blocks[i], smiles_data[i]);
            } catch (const std::exception& e) {
                errors[job] = e.what();
            }
        });
    }
    for (size_t i = 0; i < id_blocks.size(); i++, job++) {
        std::fprintf(stderr, "  loading ID %zu of %zu\n", i + 1, id_blocks.size());
        pool.emplace_back([&, i, job] {
            try {
                strings_from_block(id_blocks[i], ids_data[i]);
            } catch (const std::exception& e) {
                errors[job] = e.what();
            }
        });
    }
    for (auto& t : pool) t.join();
    for (const auto& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    for (auto& v : smiles_data) smiles_vector.insert(smiles_vector.end(), v.begin(), v.end());
    for (auto& v : ids_data) ids_vector.insert(ids_vector.end(), v.begin(), v.end());
}

} // namespace gpusim
