// Generates tests/golden/shuffle_golden.json: the permutation the reference's keyframe sampling draws
// (core/mapping/mapper.cpp:1326-1333: std::iota -> std::mt19937 g; g.seed(seed); std::shuffle(indices, g)).
// The algorithm lives in the C++ standard library, not in the reference: this program IS the reference call, compiled
// with this image's libstdc++ (g++ 11.4).   g++ -O2 -o /tmp/mk tests/golden/make_shuffle_golden.cpp && /tmp/mk > tests/golden/shuffle_golden.json
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
int main()
{
  const long seeds[] = {0, 1, 7, 1234567, 4294967301L /* > 2^32: seed() reduces it mod 2^32 */};
  const long sizes[] = {1, 2, 3, 10, 257, 4096, 16128, 20480, 65535, 65536, 70000};
  printf("{\"cases\": [\n");
  bool first = true;
  for (long seed : seeds)
    for (long n : sizes)
    {
      std::vector<long> idx(n);
      std::iota(idx.begin(), idx.end(), 0);
      std::mt19937 g;
      g.seed(seed);
      std::shuffle(idx.begin(), idx.end(), g);
      unsigned long long h = 1469598103934665603ull; // FNV-1a over the whole permutation
      for (long v : idx)
      {
        h ^= (unsigned long long)v;
        h *= 1099511628211ull;
      }
      printf("%s  {\"seed\": %ld, \"n\": %ld, \"fnv1a\": \"%llu\", \"head\": [", first ? "" : ",\n", seed, n, h);
      for (long i = 0; i < std::min(n, 16L); ++i)
        printf("%s%ld", i ? ", " : "", idx[i]);
      printf("]}");
      first = false;
    }
  printf("\n]}\n");
  return 0;
}
