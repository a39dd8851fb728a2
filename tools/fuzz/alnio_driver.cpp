#include <cstdio>
#include <cstdlib>
#include <vector>
#include "miso_alnio.h"
int main(int argc, char **argv) {
  int ok = 0, bad = 0;
  for (int a = 1; a < argc; a++) {
    miso_alnfile_t *f = nullptr;
    int rc = miso_aln_open(argv[a], 3, &f);
    if (rc) { bad++; continue; }
    ok++;
    int nref = miso_aln_n_refs(f);
    for (int r = -1; r <= nref; r++) {
      for (int rep = 0; rep < 20; rep++) {
        int64_t s = (rand() % 200000000) - 1000, e = s + (rand() % 300000);
        int64_t n = 0, nb = 0, nd = 0;
        for (int paired = 0; paired < 2; paired++) {
          miso_aln_parse_reads(f, r, s, e, paired, rep & 1, "+-?\0"[rep & 3], (rep & 4) ? 36 : 0, nullptr, 0, nullptr, 0, &n, &nb, &nd);
          std::vector<int32_t> pos(n * (paired ? 2 : 1) + 1); std::vector<char> cg(nb + 1);
          miso_aln_parse_reads(f, r, s, e, paired, rep & 1, "+-?\0"[rep & 3], (rep & 4) ? 36 : 0, pos.data(), pos.size() - 1, cg.data(), nb, &n, &nb, &nd);
        }
        std::vector<int64_t> idx(16);
        miso_aln_fetch(f, r, s, e, idx.data(), 16, &n);
      }
    }
    miso_aln_close(f);
  }
  std::printf("opened %d, rejected %d\n", ok, bad);
  return 0;
}
