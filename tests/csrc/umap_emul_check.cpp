// CPU harness: rs_umap_order (radiosaber_amd/csrc/rs_sort_emul.h, the code the gfx950 SubOpt runs to learn in which order
// the reference's `slice_fewer` hashtable yields its slices) against the real std::unordered_map<int, int>: keys inserted
// ascending, some erased afterwards, iteration order compared.  Prints "OK <cases>" or the first mismatch.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <vector>

#include "../../radiosaber_amd/csrc/rs_sort_emul.h"

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937_64 rng(4242);
  uint8_t nxt[68], bkt[128], ord[64];
  for (int c = 0; c < cases; c++) {
    uint64_t keys;
    switch (c % 6) {
      case 0: keys = rng(); break;
      case 1: keys = rng() & rng(); break;
      case 2: keys = rng() | rng(); break;
      case 3: keys = rng() & rng() & rng(); break;
      case 4: keys = c < 600 ? (c / 6 >= 64 ? ~0ull : ((1ull << (c / 6)) - 1) | (1ull << (c / 6))) : ~(rng() & rng() & rng()); break;
      default: keys = rng() & ((1ull << (1 + rng() % 63)) - 1); break;
    }
    const uint64_t erased = (c % 3 == 0) ? (rng() & rng()) : 0;
    std::unordered_map<int, int> m;
    for (int k = 0; k < 64; k++)
      if ((keys >> k) & 1) m[k] = k + 1;
    for (int k = 0; k < 64; k++)
      if (((keys & erased) >> k) & 1) m.erase(k);
    std::vector<int> want;
    for (auto it = m.begin(); it != m.end(); ++it) want.push_back(it->first);
    const int n = rs_umap_order(keys, nxt, bkt, ord);
    std::vector<int> got;
    for (int i = 0; i < n; i++)
      if (!(((keys & erased) >> ord[i]) & 1)) got.push_back(ord[i]);
    if (got != want) {
      printf("MISMATCH case %d keys %016llx erased %016llx\n want:", c, (unsigned long long)keys, (unsigned long long)erased);
      for (int v : want) printf(" %d", v);
      printf("\n got: ");
      for (int v : got) printf(" %d", v);
      printf("\n");
      return 1;
    }
  }
  printf("OK %d\n", cases);
  return 0;
}
