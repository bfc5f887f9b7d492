// fsim_reader.h -- loader for the reference's .fsim database files.
// Twin of GPUSimServer::extractData (gpusim.cpp:173-253); format in qds.h /
// SURVEY.md Appendix B.  The file is mapped, not read into the heap; zlib inflates the
// qCompress blocks where they lie, on a bounded pool of workers (the reference uses the
// global QThreadPool, :193-236), each block's bytes held once.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace gpusim
{

constexpr int DATABASE_VERSION = 3; // gpusim.cpp:43

// Throws std::runtime_error("Database version incompatible with this GPUSim
// version") on a version mismatch (gpusim.cpp:186-189) and on unreadable /
// truncated files.  smiles / ids are new[]-allocated C strings (never freed by the
// reference either: the FingerprintDB adopts them).
void extractData(const std::string& database_fname, int& fp_bitcount, int& fp_count, std::string& dbkey,
                 std::vector<std::vector<char>>& fingerprint_data, std::vector<char*>& smiles_vector,
                 std::vector<char*>& ids_vector);

// Workers extractData inflates with: min(jobs, hardware threads, kMaxExtractThreads).
constexpr unsigned kMaxExtractThreads = 32;
unsigned extract_threads(size_t jobs);

// qUncompress: u32 big-endian expected size + zlib stream.
std::vector<unsigned char> q_uncompress(const std::vector<unsigned char>& blob);

} // namespace gpusim
