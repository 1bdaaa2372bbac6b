// fa_stats.h -- host-side parameter statistics of the FastANI path and the
// lookup tables the device kernels read instead of doing floating point.
//
// Replaces skch::Stat::* (declared include/fastani/map/map_stats.pxd:6-29 of
// the reference; the C++ itself lives in the un-vendored FastANI submodule)
// and the Boost.Math binomial it uses.  All kernels are integer-only: every
// float the path produces is a pure function of (sketch size s, shared count
// c, k, identity threshold) and is tabulated here once per mapper.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace fa {

// Stat::j2md.  C++ expression: float r = (-1.0 / k) * log(2.0 * j / (1 + j)); (1 + j) is evaluated in float.
inline float stat_j2md(float j, int k) {
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  float one_plus_j = 1 + j;
  return (float)((-1.0 / k) * std::log(2.0 * (double)j / (double)one_plus_j));
}

// Stat::md2j.  float r = 1.0 / (2.0 * exp(k * d) - 1.0); k*d is float and exp resolves to the float overload in
// the pyfastani translation unit (Python.h includes <math.h>, whose C++ wrapper exports std::exp(float)).
inline float stat_md2j(float d, int k) {
  float kd = (float)k * d;
  float e = std::exp(kd);
  return (float)(1.0 / (2.0 * (double)e - 1.0));
}

// Upper quantile of Binomial(n, p): the smallest x with CDF(x) >= 1 - q, i.e. what
// boost::math::quantile(complement(binomial(n, p), q)) returns under the default integer_round_outwards policy.
inline int stat_binomial_upper_quantile(int n, double p, double q) {
  const double target = 1.0 - q;
  if (n <= 0) return 0;
  if (p >= 1.0) return n;
  if (p <= 0.0) return 0;
  const double log_p = std::log(p), log_q = std::log1p(-p);
  if (target <= std::exp(n * log_q)) return 0;
  const double lg_n = std::lgamma((double)n + 1.0);
  double cdf = 0.0;
  for (int x = 0; x < n; x++) {
    cdf += std::exp(lg_n - std::lgamma(x + 1.0) - std::lgamma((double)(n - x) + 1.0) + x * log_p + (n - x) * log_q);
    if (cdf >= target) return x;
  }
  return n;
}

// P[X >= x], X ~ Binomial(n, r)
inline double stat_binomial_tail(int n, double r, int x) {
  if (x <= 0) return 1.0;
  if (x > n || r <= 0.0) return 0.0;
  if (r >= 1.0) return 1.0;
  const double log_p = std::log(r), log_q = std::log1p(-r), lg_n = std::lgamma((double)n + 1.0);
  double sum = 0.0;
  for (int i = x; i <= n; i++) {
    double t = std::exp(lg_n - std::lgamma(i + 1.0) - std::lgamma((double)(n - i) + 1.0) + i * log_p + (n - i) * log_q);
    sum += t;
    if (t < sum * 1e-18 && i > n * r) break;
  }
  return sum;
}

const float kConfidence = 0.9f;  // confidence interval used by estimateMinimumHitsRelaxed and doL2Mapping

// Stat::md_lower_bound
inline float stat_md_lower_bound(float d, int s, int k, float ci) {
  double q2 = (1.0 - (double)ci) / 2.0;
  int x = stat_binomial_upper_quantile(s, (double)stat_md2j(d, k), q2);
  return stat_j2md((float)x / (float)s, k);
}

// nucIdentity and nucIdentityUpperBound of Map::doL2Mapping for c shared sketch elements out of s
inline void stat_identity(int c, int s, int k, float *identity, float *upper) {
  float md = stat_j2md((float)(1.0 * c / s), k);
  float lo = stat_md_lower_bound(md, s, k, kConfidence);
  *identity = 100 * (1 - md);
  *upper = 100 * (1 - lo);
}

inline int stat_min_hits(int s, int k, float pid) {
  float mash = (float)(1.0 - (double)pid / 100.0);
  float jaccard = stat_md2j(mash, k);
  return (int)std::ceil(1.0 * s * (double)jaccard);
}

// Stat::estimateMinimumHitsRelaxed
inline int stat_min_hits_relaxed(int s, int k, float pid) {
  int first = stat_min_hits(s, k, pid), relaxed = first;
  for (int i = first; i >= 0; i--) {
    float d = stat_j2md((float)(1.0 * i / s), k);
    float lo = stat_md_lower_bound(d, s, k, kConfidence);
    float upper = (float)(100.0 * (1.0 - (double)lo));
    if (upper >= pid) relaxed = i; else break;
  }
  return relaxed;
}

// Stat::estimate_pvalue
inline double stat_pvalue(int s, int k, int alphabet, float identity, int len_query, uint64_t len_ref) {
  double space = std::pow((double)alphabet, (double)k);
  double px = 1.0 / (1.0 + space / len_query), py = px;
  double r = px * py / (px + py - px * py);
  int x = stat_min_hits_relaxed(s, k, identity);
  double tail = (x == 0) ? 1.0 : stat_binomial_tail(s, r, x);
  return (double)len_ref * tail;
}

// Stat::recommendedWindowSize; -1 when no candidate sketch size reaches the cut-off
inline int stat_recommended_window(double cutoff, int k, int alphabet, float identity, int len_query, uint64_t len_ref) {
  std::vector<int> candidates{1, 2, 5};
  for (int i = 10; i < len_query; i += 10) candidates.push_back(i);
  for (int s : candidates) {
    if (stat_pvalue(s, k, alphabet, identity, len_query, len_ref) <= cutoff) {
      int w = (int)(2.0 * len_query / s);
      return std::min(std::max(w, 1), len_query);
    }
  }
  return -1;
}

// Tables indexed by sketch size s (0..smax): minimum L1 hits, the smallest shared count whose upper-bound
// identity passes the percentage_identity filter of doL2Mapping, and the triangular identity table
// ident[s*(s+1)/2 + c] = nucIdentity(c, s) as float bits.
struct StatTables {
  int k = 16;
  float pid = 80.0f;
  int smax = -1;
  std::vector<int32_t> min_hits;     // [s]
  std::vector<int32_t> pass_shared;  // [s]  (s+1 = nothing passes)
  std::vector<float> ident;          // triangular

  static size_t tri(int s) { return (size_t)s * (size_t)(s + 1) / 2; }

  bool passes(int c, int s) const {
    float id, up;
    stat_identity(c, s, k, &id, &up);
    return up >= pid;
  }

  // grows the tables to cover sketch sizes up to new_smax; returns true if anything changed
  bool extend(int new_smax) {
    if (new_smax <= smax) return false;
    min_hits.resize(new_smax + 1);
    pass_shared.resize(new_smax + 1);
    ident.resize(tri(new_smax + 1) + new_smax + 1);
    for (int s = std::max(smax + 1, 0); s <= new_smax; s++) {
      if (s == 0) { min_hits[0] = 0; pass_shared[0] = 1; ident[0] = 0.0f; continue; }
      min_hits[s] = stat_min_hits_relaxed(s, k, pid);
      // The upper-bound identity is monotone in c; locate the threshold by walking from the strict estimate.
      int c = std::min(std::max(stat_min_hits(s, k, pid), 0), s);
      if (passes(c, s)) { while (c > 0 && passes(c - 1, s)) c--; }
      else { while (c <= s && !passes(c, s)) c++; }
      pass_shared[s] = c;
      for (int j = 0; j <= s; j++) ident[tri(s) + j] = 100 * (1 - stat_j2md((float)(1.0 * j / s), k));
    }
    smax = new_smax;
    return true;
  }
};

}  // namespace fa
