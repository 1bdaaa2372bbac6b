// Host-only ingest rate (no GPU): N synthetic genomes as FASTA files in /dev/shm (60-column lines), read and 2-bit packed
//   (a) one file after another through read_fasta_records + HostStore::append_many (the round-4 path of add_fasta /
//       upload_fasta: every file spread over the pool, three sweeps),
//   (b) all at once through read_fasta_packed_many + append_packed (round 5: one task per file, one sweep).
//   g++ -O2 -std=c++17 -pthread -o ingest_host ingest_host.cpp && ./ingest_host [files] [bases]
#include <chrono>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "../../pyfastani_amd/csrc/fa_fasta.h"

using namespace fa;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200;
  const size_t len = argc > 2 ? (size_t)atoll(argv[2]) : 5000000;
  const std::string dir = std::string("/dev/shm/fa_ingest_") + std::to_string((int)getpid());
  if (system(("mkdir -p " + dir).c_str()) != 0) return 1;
  std::vector<std::string> paths;
  size_t bytes = 0;
  {
    std::mt19937_64 rng(7);
    std::string text;
    text.reserve(len + len / 60 + 64);
    for (int i = 0; i < n; i++) {
      text.assign(">genome_" + std::to_string(i) + "\n");
      // (one random genome, rotated per file: generation is not what is measured)
      static std::string body;
      if (body.empty()) { body.resize(len); for (size_t k = 0; k < len; k++) body[k] = "ACGT"[rng() & 3]; }
      const size_t rot = (size_t)(rng() % len);
      for (size_t k = 0; k < len; k += 60) {
        const size_t m = std::min<size_t>(60, len - k), a = (k + rot) % len;
        if (a + m <= len) text.append(body, a, m); else { text.append(body, a, len - a); text.append(body, 0, m - (len - a)); }
        text += '\n';
      }
      const std::string p = dir + "/g" + std::to_string(i) + ".fna";
      FILE *f = fopen(p.c_str(), "wb");
      if (!f || fwrite(text.data(), 1, text.size(), f) != text.size()) { fprintf(stderr, "cannot write %s\n", p.c_str()); return 1; }
      fclose(f);
      paths.push_back(p);
      bytes += text.size();
    }
  }
  std::vector<const char *> cp;
  for (auto &p : paths) cp.push_back(p.c_str());
  HostPool::get();
  double t_old = 1e30, t_new = 1e30, t_read = 1e30, t_new_first = 0;
  uint64_t sum_old = 0, sum_new = 0;
  for (int rep = 0; rep < 3; rep++) {
    {
      const double t0 = now();
      HostStore hs;
      for (int i = 0; i < n; i++) {
        std::vector<FastaSeq> seqs;
        read_fasta_records(cp[i], seqs);
        std::vector<const void *> ptrs; std::vector<int64_t> lens;
        for (auto &q : seqs) { ptrs.push_back(q.data.get()); lens.push_back((int64_t)q.size); }
        hs.append_many(ptrs.data(), lens.data(), (int64_t)ptrs.size(), 1);
      }
      t_old = std::min(t_old, now() - t0);
      sum_old = 0; for (uint32_t w : hs.packed) sum_old = sum_old * 1000003ULL + w;
    }
    {
      const double t0 = now();
      std::vector<PackedFasta> files;
      read_fasta_packed_many(cp.data(), cp.size(), false, files);
      const double t1 = now();
      HostStore hs;
      std::vector<PackedRef> refs;
      for (auto &f : files) for (size_t r = 0; r < f.rec_len.size(); r++) refs.push_back(PackedRef{&f, (int64_t)r, f.rec_len[r]});
      append_packed(hs, refs.data(), (int64_t)refs.size());
      const double t2 = now();
      if (rep == 0) t_new_first = t2 - t0;                            // (the first call of the process: cold allocator, cold threads)
      if (t2 - t0 < t_new) { t_new = t2 - t0; t_read = t1 - t0; }
      sum_new = 0; for (uint32_t w : hs.packed) sum_new = sum_new * 1000003ULL + w;
    }
  }
  for (auto &p : paths) unlink(p.c_str());
  rmdir(dir.c_str());
  printf("{\"files\": %d, \"bytes\": %zu, \"host_threads\": %d, \"one_by_one_s\": %.4f, \"one_by_one_GBps\": %.2f, \"concurrent_s\": %.4f, \"concurrent_GBps\": %.2f, "
         "\"concurrent_read_pack_s\": %.4f, \"concurrent_place_s\": %.4f, \"concurrent_first_call_s\": %.4f, \"concurrent_first_call_GBps\": %.2f, \"stores_equal\": %s}\n",
         n, bytes, host_threads(), t_old, bytes / t_old / 1e9, t_new, bytes / t_new / 1e9, t_read, t_new - t_read, t_new_first, bytes / t_new_first / 1e9,
         sum_old == sum_new ? "true" : "false");
  return sum_old == sum_new ? 0 : 1;
}
