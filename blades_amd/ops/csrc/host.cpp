// Host-side native runtime pieces (plain C ABI, loaded through ctypes).
//
// The reference leans on scikit-learn's Cython/C++ (KMeans, Agglomerative
// clustering: fllib/aggregators/signguard.py:59, clippedclustering.py:53-56),
// numpy's C random/partition routines (fllib/datasets/fldataset.py:169-214) and
// Ray's C++ scheduler for placing clients on workers.  These are the native
// equivalents this framework owns.  K (clients) is at most a few thousand, so
// everything here is O(K^2) on one core and runs in well under a millisecond
// for the shipped 60-client configs (≈10 ms at K = 1000).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <queue>
#include <random>
#include <vector>

namespace {

struct Merge {
  int a, b;       // cluster representatives (original leaf index of each side)
  double height;  // linkage distance at the merge
  int order;      // creation order, tie-break for a stable sort
};

struct DSU {
  std::vector<int> p;
  explicit DSU(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
  int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
  void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

// Nearest-neighbour-chain agglomerative clustering on a dense distance matrix
// with Lance–Williams updates.  linkage: 0 = single, 1 = average, 2 = complete.
// All three are reducible, so the chain finds exactly the merges the greedy
// "closest pair first" algorithm makes; sorting them by height afterwards
// recovers the dendrogram order.
void nn_chain(std::vector<double>& D, int n, int linkage, std::vector<Merge>& merges) {
  std::vector<int> size(n, 1);
  std::vector<char> alive(n, 1);
  std::vector<int> chain;
  chain.reserve(n);
  int remaining = n;
  auto at = [&](int i, int j) -> double& { return D[(size_t)i * n + j]; };
  while (remaining > 1) {
    if (chain.empty()) {
      for (int i = 0; i < n; ++i) if (alive[i]) { chain.push_back(i); break; }
    }
    while (true) {
      int x = chain.back();
      int prev = chain.size() >= 2 ? chain[chain.size() - 2] : -1;
      int best = -1;
      double bd = std::numeric_limits<double>::infinity();
      // prefer the previous chain element on ties so the chain terminates
      if (prev >= 0) { best = prev; bd = at(x, prev); }
      for (int j = 0; j < n; ++j) {
        if (!alive[j] || j == x) continue;
        double d = at(x, j);
        if (d < bd) { bd = d; best = j; }
      }
      if (best == prev && prev >= 0) {
        // reciprocal nearest neighbours: merge x and prev
        chain.pop_back();
        chain.pop_back();
        int a = std::min(x, prev), b = std::max(x, prev);
        merges.push_back({a, b, bd, (int)merges.size()});
        int sa = size[a], sb = size[b];
        for (int k = 0; k < n; ++k) {
          if (!alive[k] || k == a || k == b) continue;
          double da = at(a, k), db = at(b, k), nd;
          if (linkage == 0) nd = std::min(da, db);
          else if (linkage == 2) nd = std::max(da, db);
          else nd = (sa * da + sb * db) / (double)(sa + sb);
          at(a, k) = nd;
          at(k, a) = nd;
        }
        alive[b] = 0;
        size[a] = sa + sb;
        --remaining;
        break;
      }
      chain.push_back(best);
    }
  }
}

}  // namespace

extern "C" {

// Two-cluster cut of the agglomerative dendrogram of an n x n precomputed
// distance matrix (row-major doubles).  labels[i] in {0,1}; the cluster that
// contains point 0 ... is labelled so that label 0 is the cluster holding the
// smallest index (deterministic; sklearn's numbering only matters in an exact
// size tie, see Clippedclustering).  Returns 0 on success.
int blades_agglomerative2(const double* dist, int n, int linkage, int* labels) {
  if (n <= 0) return 1;
  if (n == 1) { labels[0] = 0; return 0; }
  std::vector<double> D(dist, dist + (size_t)n * n);
  std::vector<Merge> merges;
  merges.reserve(n - 1);
  nn_chain(D, n, linkage, merges);
  std::stable_sort(merges.begin(), merges.end(), [](const Merge& x, const Merge& y) {
    return x.height < y.height;
  });
  DSU dsu(n);
  for (size_t m = 0; m + 1 < merges.size(); ++m) dsu.unite(merges[m].a, merges[m].b);
  int root0 = dsu.find(0);
  for (int i = 0; i < n; ++i) labels[i] = dsu.find(i) == root0 ? 0 : 1;
  return 0;
}

// Lloyd k-means with k-means++ seeding, n_init restarts, best inertia wins.
// X is n x d row-major doubles; labels out [n]; centers out [k x d].
int blades_kmeans(const double* X, int n, int d, int k, uint64_t seed, int n_init, int max_iter,
                  int* labels, double* centers) {
  if (n <= 0 || k <= 0) return 1;
  std::mt19937_64 rng(seed);
  std::vector<double> best_c((size_t)k * d);
  std::vector<int> best_l(n, 0);
  double best_inertia = std::numeric_limits<double>::infinity();
  std::vector<double> c((size_t)k * d), mind(n), sum((size_t)k * d);
  std::vector<int> l(n), cnt(k);
  auto sq = [&](const double* a, const double* b) {
    double s = 0;
    for (int t = 0; t < d; ++t) { double z = a[t] - b[t]; s += z * z; }
    return s;
  };
  for (int init = 0; init < std::max(1, n_init); ++init) {
    // k-means++ seeding
    std::uniform_int_distribution<int> pick(0, n - 1);
    int first = pick(rng);
    std::memcpy(&c[0], X + (size_t)first * d, sizeof(double) * d);
    for (int i = 0; i < n; ++i) mind[i] = sq(X + (size_t)i * d, &c[0]);
    for (int j = 1; j < k; ++j) {
      double tot = std::accumulate(mind.begin(), mind.end(), 0.0);
      int chosen = 0;
      if (tot > 0) {
        std::uniform_real_distribution<double> u(0.0, tot);
        double r = u(rng), acc = 0;
        for (int i = 0; i < n; ++i) { acc += mind[i]; if (acc >= r) { chosen = i; break; } }
      } else {
        chosen = pick(rng);
      }
      std::memcpy(&c[(size_t)j * d], X + (size_t)chosen * d, sizeof(double) * d);
      for (int i = 0; i < n; ++i) mind[i] = std::min(mind[i], sq(X + (size_t)i * d, &c[(size_t)j * d]));
    }
    double inertia = 0;
    for (int it = 0; it < max_iter; ++it) {
      bool changed = false;
      inertia = 0;
      for (int i = 0; i < n; ++i) {
        int bj = 0;
        double bd = std::numeric_limits<double>::infinity();
        for (int j = 0; j < k; ++j) {
          double dd = sq(X + (size_t)i * d, &c[(size_t)j * d]);
          if (dd < bd) { bd = dd; bj = j; }
        }
        if (it == 0 || l[i] != bj) changed = true;
        l[i] = bj;
        inertia += bd;
      }
      if (!changed) break;
      std::fill(sum.begin(), sum.end(), 0.0);
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int i = 0; i < n; ++i) {
        ++cnt[l[i]];
        for (int t = 0; t < d; ++t) sum[(size_t)l[i] * d + t] += X[(size_t)i * d + t];
      }
      for (int j = 0; j < k; ++j)
        if (cnt[j] > 0)
          for (int t = 0; t < d; ++t) c[(size_t)j * d + t] = sum[(size_t)j * d + t] / cnt[j];
    }
    if (inertia < best_inertia) {
      best_inertia = inertia;
      best_c = c;
      best_l = l;
    }
  }
  // canonical numbering: the cluster of point 0 is label 0
  if (k == 2 && best_l[0] == 1) {
    for (int i = 0; i < n; ++i) best_l[i] ^= 1;
    for (int t = 0; t < d; ++t) std::swap(best_c[t], best_c[(size_t)d + t]);
  }
  std::memcpy(labels, best_l.data(), sizeof(int) * n);
  if (centers) std::memcpy(centers, best_c.data(), sizeof(double) * k * d);
  return 0;
}

// Non-IID label partition: per class, shuffle its sample ids, draw
// Dirichlet(alpha) client proportions, zero the share of clients that already
// hold >= N/num_clients samples, split.  Retries until every client has at
// least min_size samples (algorithm of fllib/datasets/fldataset.py:180-214).
// out_client[i] = client of sample i.  Returns number of attempts (<0: error).
int blades_dirichlet_partition(const int64_t* y, int64_t N, int num_classes, int num_clients,
                               double alpha, uint64_t seed, int min_size, int32_t* out_client) {
  if (N <= 0 || num_clients <= 0) return -1;
  std::mt19937_64 rng(seed);
  std::vector<std::vector<int64_t>> by_class(num_classes);
  for (int64_t i = 0; i < N; ++i) {
    if (y[i] < 0 || y[i] >= num_classes) return -2;
    by_class[y[i]].push_back(i);
  }
  std::gamma_distribution<double> gam(alpha, 1.0);
  for (int attempt = 1; attempt <= 1000; ++attempt) {
    std::vector<int64_t> count(num_clients, 0);
    std::vector<int32_t> assign(N, -1);
    for (int k = 0; k < num_classes; ++k) {
      std::vector<int64_t> idx = by_class[k];
      std::shuffle(idx.begin(), idx.end(), rng);
      std::vector<double> p(num_clients);
      double tot = 0;
      for (int j = 0; j < num_clients; ++j) {
        p[j] = gam(rng);
        if ((double)count[j] >= (double)N / num_clients) p[j] = 0;
        tot += p[j];
      }
      if (tot <= 0) { for (auto& v : p) v = 1.0; tot = num_clients; }
      double acc = 0;
      int64_t start = 0;
      for (int j = 0; j < num_clients; ++j) {
        acc += p[j] / tot;
        int64_t end = (j == num_clients - 1) ? (int64_t)idx.size() : (int64_t)(acc * idx.size());
        end = std::min<int64_t>(std::max(end, start), idx.size());
        for (int64_t t = start; t < end; ++t) assign[idx[t]] = j;
        count[j] += end - start;
        start = end;
      }
    }
    int64_t mn = *std::min_element(count.begin(), count.end());
    if (mn >= min_size) {
      std::memcpy(out_client, assign.data(), sizeof(int32_t) * N);
      return attempt;
    }
  }
  return -3;
}

// Longest-processing-time placement of clients on ranks: cost[i] -> rank with
// the least load so far, heaviest first.  Replaces Ray's affinity actor pool
// (fllib/core/execution/actor_manager.py:48-81) with a static, balanced map.
int blades_lpt_assign(const double* cost, int n, int world, int32_t* out_rank) {
  if (world <= 0) return 1;
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  using P = std::pair<double, int>;
  std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
  for (int r = 0; r < world; ++r) pq.push({0.0, r});
  for (int i : order) {
    P top = pq.top();
    pq.pop();
    out_rank[i] = top.second;
    pq.push({top.first + cost[i], top.second});
  }
  return 0;
}

}  // extern "C"
