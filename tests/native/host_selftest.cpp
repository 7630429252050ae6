// Self-test of the host-side native code (ops/csrc/host.cpp, parallel/csrc/launch.cpp helpers excluded), meant to be
// built with -fsanitize=address,undefined: exercises every exported entry point on edge-case sizes so that
// out-of-bounds accesses, leaks and undefined arithmetic in the clustering / partitioning code surface on the CPU.
// (GPU sanitizers are not available on the target pool; SURVEY.md 5.2 asks for a sanitized host build.)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

extern "C" {
int blades_agglomerative2(const double* dist, int n, int linkage, int* labels);
int blades_kmeans(const double* X, int n, int d, int k, uint64_t seed, int n_init, int max_iter, int* labels,
                  double* centers);
int blades_dirichlet_partition(const int64_t* y, int64_t N, int num_classes, int num_clients, double alpha,
                               uint64_t seed, int min_size, int32_t* out_client);
int blades_lpt_assign(const double* cost, int n, int world, int32_t* out_rank);
}

#define CHECK(c)                                                          \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                       \
    }                                                                     \
  } while (0)

static void test_agglomerative() {
  std::mt19937_64 rng(5);
  std::normal_distribution<double> nd(0.0, 0.05);
  for (int n : {1, 2, 3, 7, 64, 257}) {
    // two well separated groups on a line: [0, n/3) near 0, the rest near 10
    std::vector<double> pos(n);
    int cut = n / 3;
    for (int i = 0; i < n; ++i) pos[i] = (i < cut ? 0.0 : 10.0) + nd(rng);
    std::vector<double> D((size_t)n * n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) D[(size_t)i * n + j] = std::fabs(pos[i] - pos[j]);
    for (int linkage = 0; linkage < 3; ++linkage) {
      std::vector<int> lab(n, -1);
      int rc = blades_agglomerative2(D.data(), n, linkage, lab.data());
      if (n < 2) continue;  // nothing to split; must simply not touch memory it does not own
      CHECK(rc == 0);
      for (int i = 0; i < n; ++i) CHECK(lab[i] == 0 || lab[i] == 1);
      if (cut > 0)
        for (int i = 0; i < n; ++i) CHECK((lab[i] == lab[0]) == (i < cut));
    }
  }
  // NaN / inf entries as the aggregators sanitise them (2 / 0) must not break the merge loop
  std::vector<double> D = {0, 2, 0, 2, 0, 2, 0, 2, 0};
  std::vector<int> lab(3);
  CHECK(blades_agglomerative2(D.data(), 3, 1, lab.data()) == 0);
}

static void test_kmeans() {
  std::mt19937_64 rng(9);
  std::normal_distribution<double> nd(0.0, 0.01);
  for (int n : {1, 2, 5, 100}) {
    for (int d : {1, 3}) {
      std::vector<double> X((size_t)n * d);
      for (int i = 0; i < n; ++i)
        for (int t = 0; t < d; ++t) X[(size_t)i * d + t] = (i % 2 ? 1.0 : 0.0) + nd(rng);
      std::vector<int> lab(n, -1);
      std::vector<double> cen((size_t)2 * d);
      int rc = blades_kmeans(X.data(), n, d, 2, 0, 3, 50, lab.data(), cen.data());
      CHECK(rc == 0);
      for (int i = 0; i < n; ++i) CHECK(lab[i] == 0 || lab[i] == 1);
      if (n >= 4)
        for (int i = 2; i < n; ++i) CHECK(lab[i] == lab[i % 2]);
    }
  }
  // all points identical: k-means++ has zero total distance to draw from
  std::vector<double> X(12, 0.5);
  std::vector<int> lab(4);
  std::vector<double> cen(6);
  CHECK(blades_kmeans(X.data(), 4, 3, 2, 1, 2, 10, lab.data(), cen.data()) == 0);
  CHECK(blades_kmeans(X.data(), 0, 3, 2, 1, 2, 10, lab.data(), cen.data()) != 0);
}

static void test_partition() {
  const int64_t N = 3000;
  std::vector<int64_t> y(N);
  for (int64_t i = 0; i < N; ++i) y[i] = i % 10;
  for (double alpha : {0.1, 1.0, 100.0}) {
    std::vector<int32_t> out(N, -1);
    int attempts = blades_dirichlet_partition(y.data(), N, 10, 12, alpha, 7, 10, out.data());
    CHECK(attempts >= 1);
    std::vector<int> cnt(12, 0);
    for (int64_t i = 0; i < N; ++i) {
      CHECK(out[i] >= 0 && out[i] < 12);
      cnt[out[i]]++;
    }
    for (int c : cnt) CHECK(c >= 10);
  }
  std::vector<int32_t> out(N);
  y[17] = 11;  // label outside [0, num_classes)
  CHECK(blades_dirichlet_partition(y.data(), N, 10, 12, 1.0, 7, 10, out.data()) == -2);
  CHECK(blades_dirichlet_partition(y.data(), 0, 10, 12, 1.0, 7, 10, out.data()) == -1);
}

static void test_lpt() {
  std::vector<double> cost = {5, 1, 1, 1, 1, 1, 4, 4};
  std::vector<int32_t> r(cost.size(), -1);
  CHECK(blades_lpt_assign(cost.data(), (int)cost.size(), 3, r.data()) == 0);
  std::vector<double> load(3, 0);
  for (size_t i = 0; i < cost.size(); ++i) {
    CHECK(r[i] >= 0 && r[i] < 3);
    load[r[i]] += cost[i];
  }
  for (double l : load) CHECK(l >= 5.0 && l <= 7.0);
  CHECK(blades_lpt_assign(cost.data(), 0, 3, r.data()) == 0);
  CHECK(blades_lpt_assign(cost.data(), 3, 0, r.data()) != 0);
}

int main() {
  test_agglomerative();
  test_kmeans();
  test_partition();
  test_lpt();
  std::puts("HOST_SELFTEST_OK");
  return 0;
}
