// Driver of the host helper-thread pool of pk_runtime.cpp (pk_host_threads / pk_same_bits / pk_copy_bits) for the
// ThreadSanitizer build (tests/test_runtime_sanitized.py): passes of different lengths, helpers hot and cold, differences at
// the front, in the middle and at the very end.  CPU only, no HIP call is made.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "include/pockit_hip.h"

#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) {                                                               \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                              \
    }                                                                            \
  } while (0)

int main() {
  std::mt19937_64 rng(11);
  long checks = 0;
  for (int helpers : {0, 1, 3, 5}) {
    CHECK(pk_host_threads(helpers) == 0);
    for (size_t n : {(size_t)1, (size_t)511, (size_t)131072, (size_t)131073, (size_t)400001, (size_t)1000003}) {
      std::vector<double> a(n), b(n), c(n);
      for (auto& v : a) v = (double)(rng() % 1000003) * 0.5;
      for (int rep = 0; rep < 12; ++rep) {
        if (rep % 4 == 3) std::this_thread::sleep_for(std::chrono::milliseconds(3));      // helpers go cold
        CHECK(pk_copy_bits(b.data(), a.data(), n) == 0);
        CHECK(pk_same_bits(a.data(), b.data(), n) == 1);
        const size_t where = rep % 3 == 0 ? n - 1 : (rep % 3 == 1 ? (size_t)(rng() % n) : 0);
        const double keep = b[where];
        b[where] = keep + 1.0;
        CHECK(pk_same_bits(a.data(), b.data(), n) == 0);
        b[where] = keep;
        CHECK(pk_same_bits(b.data(), a.data(), n) == 1);
        for (size_t i = 0; i < n; i += 4099) c[i] = -1.0;
        CHECK(pk_copy_bits(c.data(), b.data(), n) == 0);
        for (size_t i = 0; i < n; i += 977) CHECK(c[i] == a[i]);
        CHECK(c[n - 1] == a[n - 1]);
        checks += 6;
      }
    }
  }
  CHECK(pk_host_threads(17) != 0);
  CHECK(pk_host_threads(0) == 0);
  std::printf("host pool driver: %ld checks passed\n", checks);
  return 0;
}
