// sxmc_plan.h -- the HOST PLANNERS of libsxmc_hip.so: every table the gfx950 kernels index blindly is laid out by
// one of these functions, from plain vectors into plain vectors, with no HIP call and no library state.  sxmc_launch_plan.cpp
// calls them and uploads what they return; tests/cpp/test_plan.cpp calls the same functions with randomized shapes
// under AddressSanitizer + UndefinedBehaviorSanitizer, replays the kernels' addressing against the results and
// checks that nothing is missed, counted twice or addressed out of range -- without a device.
//
//   apportion_workgroups / interleaved_segments / build_partition   who reads which units (SxSegment lists)
//   eval_point_bins        EvalHist::SetEvalPoints' loop, src/pdfz.cpp:264-301
//   build_sparse_tables    sparse counting (histograms beyond LDS): bit filters + open-addressing table
//   bucket_granules / bucket_key_offsets / bucketed_layout   bucketed copies of a sample table (layout_kernels.hip)
//   bucket_tables          per-bucket event-bin tables of fill_sparse_kernel
//   event_classes          distinct tuples of event bins, weighted by multiplicity (eval_nll_kernel)
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

#include "sxmc_device_types.h"

namespace sxplan {

inline int ceil_log2(size_t x) {
  int b = 0;
  while (((size_t)1 << b) < x) b++;
  return b;
}

// ---------------------------------------------------------------------------------------------- work partition
// Cut a launch's work (units of SXMC_VEC samples) into per-workgroup segment lists.
//  interleaved: member j gets K_j workgroups (proportional to its size, sum = grid); workgroup i of
//    the member takes chunks i, i + K_j, ... of `threads` units: neighbouring workgroups read
//    neighbouring chunks at the same time and every workgroup flushes one histogram once.
//  sliced: workgroup b owns the contiguous slice [total*b/G, total*(b+1)/G) of the concatenated
//    members: perfectly balanced, used when members outnumber workgroups or are tiny.
// Largest-remainder apportionment of `grid` workgroups over members of `sizes` units: at least one per
// non-empty member and never more than the member has chunks of `threads` units.  false: they do not fit.
inline bool apportion_workgroups(const std::vector<unsigned long long>& sizes, int grid, int threads,
                                 std::vector<int>& K) {
  K.assign(sizes.size(), 0);
  unsigned long long total = 0;
  for (unsigned long long n : sizes) total += n;
  if (total == 0) return true;
  int used = 0;
  std::vector<std::pair<double, int>> frac;
  for (size_t j = 0; j < sizes.size(); j++) {
    if (!sizes[j]) continue;
    const double share = (double)grid * (double)sizes[j] / (double)total;
    const unsigned long long chunks = (sizes[j] + threads - 1) / threads;
    K[j] = (int)std::min<unsigned long long>(chunks, std::max<unsigned long long>(1, (unsigned long long)share));
    used += K[j];
    frac.push_back({share - std::floor(share), (int)j});
  }
  std::sort(frac.begin(), frac.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  for (size_t i = 0; used < grid && !frac.empty() && i < 4 * frac.size(); i++) {
    const int j = frac[i % frac.size()].second;
    const unsigned long long chunks = (sizes[(size_t)j] + threads - 1) / threads;
    if ((unsigned long long)K[(size_t)j] < chunks) {
      K[(size_t)j]++;
      used++;
    }
  }
  return used <= grid;
}

// Workgroup i of member j takes chunks i, i + K_j, ... of `threads` units of the member (nvec[j] units).
// groups > 1 (tables whose rows are SORTED by bin -- bucketed copies): the member's workgroups are split into `groups`
// teams, team t takes a contiguous part of the member (sized by its share of the workgroups, cut at multiples of
// `threads` units) and interleaves inside it.  Neighbouring workgroups of a team still read neighbouring chunks at
// the same time (DRAM locality), but a workgroup now sees only its team's part of the sorted order, i.e. a fraction
// of the histogram's bins -- and the flush sends one memory-side atomic per NON-ZERO bin of every workgroup.
inline void interleaved_segments(const std::vector<unsigned long long>& nvec, const std::vector<int>& K, int threads,
                                 int grid, std::vector<SxSegment>& segs, std::vector<unsigned>& blk_off,
                                 int groups = 1) {
  segs.clear();
  blk_off.assign(1, 0u);
  for (size_t j = 0; j < nvec.size(); j++) {
    const int Kj = K[j];
    if (Kj <= 0) continue;
    const int G = std::max(1, std::min(groups, Kj));
    const unsigned long long chunks = (nvec[j] + (unsigned long long)threads - 1) / (unsigned long long)threads;
    int wg0 = 0;
    unsigned long long chunk0 = 0;
    for (int t = 0; t < G; t++) {
      const int wg1 = (int)((long long)Kj * (t + 1) / G);                     // workgroups [wg0, wg1) form team t
      const unsigned long long chunk1 = t + 1 == G ? chunks : chunks * (unsigned long long)wg1 / (unsigned long long)Kj;
      const int Kt = wg1 - wg0;
      for (int i = 0; i < Kt; i++) {
        SxSegment sg{};
        sg.sig = (int)j;
        sg.v0 = (chunk0 + (unsigned long long)i) * (unsigned long long)threads;
        sg.v1 = std::min(nvec[j], chunk1 * (unsigned long long)threads);
        sg.step = (unsigned long long)Kt * (unsigned long long)threads;
        // (a team with more workgroups than chunks: the spare ones get nothing -- an empty list, not an empty segment,
        // because the kernels clamp loads to v1 - 1)
        if (sg.v0 < sg.v1) segs.push_back(sg);
        blk_off.push_back((unsigned)segs.size());
      }
      wg0 = wg1;
      chunk0 = chunk1;
    }
  }
  while ((int)blk_off.size() < grid + 1) blk_off.push_back((unsigned)segs.size());  // idle workgroups
}

// want_mode: 0 auto, 1 sliced, 2 interleaved.  mode_out: what was built.  align: slices start on multiples of
// `align` units (bucketed tables: granule boundaries; every member's unit count is then a multiple of it).
inline void build_partition(const std::vector<unsigned long long>& nvec, int grid, int threads, int want_mode,
                            std::vector<SxSegment>& segs, std::vector<unsigned>& blk_off, int& mode_out,
                            unsigned long long align = 1, int groups = 1) {
  segs.clear();
  blk_off.assign(1, 0u);
  unsigned long long total = 0;
  int nonempty = 0;
  for (unsigned long long n : nvec) {
    total += n;
    if (n) nonempty++;
  }
  bool interleave = want_mode == 2 || (want_mode == 0 && nonempty > 0 && grid >= 2 * nonempty &&
                                       total >= (unsigned long long)grid * threads * 8ull);
  if (want_mode == 2 && (nonempty == 0 || grid < nonempty)) interleave = false;
  std::vector<int> K;
  if (interleave) interleave = apportion_workgroups(nvec, grid, threads, K);  // false: more members than workgroups
  if (interleave) {
    interleaved_segments(nvec, K, threads, grid, segs, blk_off, groups);
    mode_out = 2;
    return;
  }
  for (int b = 0; b < grid; b++) {
    unsigned long long r0 = total * (unsigned long long)b / grid / align * align;
    const unsigned long long r1 = b + 1 == grid ? total : total * (unsigned long long)(b + 1) / grid / align * align;
    unsigned long long start = 0;
    for (size_t j = 0; j < nvec.size() && r0 < r1; j++) {
      const unsigned long long s0 = start, s1 = start + nvec[j];
      start = s1;
      if (s1 <= r0) continue;
      SxSegment sg{};
      sg.sig = (int)j;
      sg.v0 = r0 - s0;
      sg.v1 = std::min(r1, s1) - s0;
      sg.step = (unsigned long long)threads;
      segs.push_back(sg);
      r0 = std::min(r1, s1);
    }
    blk_off.push_back((unsigned)segs.size());
  }
  mode_out = 1;
}

// ---------------------------------------------------------------------------------------------- evaluation points
// EvalHist::SetEvalPoints (pdfz.cpp:264-301): the flat bin of every evaluation point, -1 outside the domain (NaN
// coordinates included; also an index that rounds up past the end, where the reference would read one past the end
// in eval_pdf), -2 for a point of another data set.  points: rows of D + 1 floats.
inline void eval_point_bins(const float* points, size_t n, int D, const double* lower, const double* upper,
                            const double* scale, const int* stride, int total_nbins, unsigned dataset,
                            std::vector<int>& rb) {
  const size_t row = (size_t)D + 1;
  rb.resize(n);
  for (size_t ip = 0; ip < n; ip++) {
    bool in_domain = true;
    int bin_id = 0;
    for (int k = 0; k < D; k++) {
      const double element = points[row * ip + (size_t)k];
      if (!(element >= lower[k] && element < upper[k])) {
        in_domain = false;
        break;
      }
      bin_id += (int)((element - lower[k]) * scale[k]) * stride[k];
    }
    if (in_domain && (unsigned)bin_id >= (unsigned)total_nbins) in_domain = false;
    if (points[row * ip + (size_t)D] != (float)dataset) bin_id = -2;  // pdfz.cpp:289-293
    rb[ip] = in_domain ? bin_id : -1;
  }
}

// ---------------------------------------------------------------------------------------------- sparse counting
// Sparse-counting structures of one evaluator from its event bins (once per SetEvalPoints): the sorted distinct
// event bins ("targets"; a bin's rank is its counter slot), every point's slot, a fine and a coarse bit filter
// and an open-addressing table {bin, slot} -- hashed exactly as fill_kernels.inc.h looks them up.
struct SparseTables {
  std::vector<unsigned> targets;
  std::vector<int> slot;                   // per evaluation point: counter slot, or the point's negative code
  std::vector<unsigned> coarse, filter, table;
  int cbits = 0, fbits = 0, tbits = 0;     // the kernels shift hashes right by 32 - bits
};

inline void build_sparse_tables(const std::vector<int>& rb, SparseTables& o) {
  o.targets.clear();
  for (int b : rb)
    if (b >= 0) o.targets.push_back((unsigned)b);
  std::sort(o.targets.begin(), o.targets.end());
  o.targets.erase(std::unique(o.targets.begin(), o.targets.end()), o.targets.end());
  const size_t T = o.targets.size();
  o.slot.resize(rb.size());
  for (size_t i = 0; i < rb.size(); i++) {
    o.slot[i] = rb[i] < 0 ? rb[i]
                          : (int)(std::lower_bound(o.targets.begin(), o.targets.end(), (unsigned)rb[i]) - o.targets.begin());
  }
  o.fbits = std::min(26, std::max(16, ceil_log2(64 * std::max<size_t>(T, 1))));  // <= 1.6 % false positives
  o.tbits = std::max(6, ceil_log2(2 * std::max<size_t>(T, 1)));                   // load <= 50 %
  // coarse filter: two hashes per bin, >= 8 bits per bin up to 2^20 bits (128 KiB of LDS): ~3 % false
  // positives at 1e5 event bins (one hash in 64 KiB let 19 % through, and every survivor costs L2 probes)
  o.cbits = std::min(20, std::max(10, ceil_log2(16 * std::max<size_t>(T, 1))));
  o.coarse.assign((size_t)1 << (o.cbits - 5), 0u);
  o.filter.assign((size_t)1 << (o.fbits - 5), 0u);
  o.table.assign((size_t)2 << o.tbits, 0xFFFFFFFFu);
  const unsigned mask = (1u << o.tbits) - 1u;
  for (size_t t = 0; t < T; t++) {
    const unsigned bin = o.targets[t];
    const unsigned hb = (bin * 0x9E3779B1u) >> (32 - o.fbits);
    o.filter[hb >> 5] |= 1u << (hb & 31u);
    const unsigned hc = (bin * 0xC2B2AE35u) >> (32 - o.cbits), hd = (bin * 0x27D4EB2Fu) >> (32 - o.cbits);
    o.coarse[hc >> 5] |= 1u << (hc & 31u);
    o.coarse[hd >> 5] |= 1u << (hd & 31u);
    unsigned hp = (bin * 0x85EBCA6Bu) >> (32 - o.tbits);
    while (o.table[2 * (size_t)hp] != 0xFFFFFFFFu) hp = (hp + 1u) & mask;
    o.table[2 * (size_t)hp] = bin;
    o.table[2 * (size_t)hp + 1] = (unsigned)t;
  }
}

// ---------------------------------------------------------------------------------------------- bucketed tables
// A table's rows sorted by bucket key (the mixed-radix tuple of the untouched observables' bin indices).
// first[k] = position of key k's first row in the sorted order (0xFFFFFFFF: no such row), first[outside] = where
// the rows outside the domain of an untouched observable start (they sort last and are dropped).  The kept rows are
// cut into LOGICAL GRANULES of at most 256 rows, bucket by bucket, each bucket padded to whole granules.
struct GranulePlan {
  std::vector<unsigned> present;           // bucket keys that have rows, ascending
  std::vector<unsigned> lsrc, lvalid, lwhich;   // per granule: first row in the sorted order, rows, index in `present`
  size_t kept = 0;                         // rows kept (inside the domain of every untouched observable)
  bool worth_it = false;                   // false: mostly padding -- the table stays in row order
};

inline void bucket_granules(const std::vector<unsigned>& first, unsigned outside, size_t n, GranulePlan& g) {
  g = GranulePlan();
  g.kept = first[outside] != 0xFFFFFFFFu ? first[outside] : n;
  for (unsigned k = 0; k < outside; k++)
    if (first[k] != 0xFFFFFFFFu) g.present.push_back(k);
  for (size_t i = 0; i < g.present.size(); i++) {
    const size_t lo = first[g.present[i]], hi = i + 1 < g.present.size() ? first[g.present[i + 1]] : g.kept;
    for (size_t at = lo; at < hi; at += 256) {
      g.lsrc.push_back((unsigned)at);
      g.lvalid.push_back((unsigned)std::min<size_t>(256, hi - at));
      g.lwhich.push_back((unsigned)i);
    }
  }
  g.worth_it = !((double)g.lsrc.size() * 256.0 > 1.3 * (double)g.kept + 16384.0);
  if (!g.worth_it) {
    g.lsrc.clear();
    g.lvalid.clear();
    g.lwhich.clear();
  }
}

// The constant contribution of a bucket to the flat bin index: sum over the untouched observables (mask) of
// index * stride, the index decoded from the key with the bases nbins + 1 (radix[k] = product of the bases of the
// untouched observables after k).
inline void bucket_key_offsets(const std::vector<unsigned>& present, unsigned mask, const unsigned* radix,
                               const int* nbins, const int* stride, int nobs, std::vector<unsigned>& key_pre) {
  key_pre.resize(present.size());
  for (size_t i = 0; i < present.size(); i++) {
    long long pre = 0;
    for (int k = 0; k < nobs; k++) {
      if (!((mask >> k) & 1u)) continue;
      const unsigned idx = (present[i] / radix[k]) % ((unsigned)nbins[k] + 1u);
      pre += (long long)idx * stride[k];
    }
    key_pre[i] = (unsigned)pre;
  }
}

// The PHYSICAL granule order of a bucketed copy laid out for `runs` runs: run r holds logical granules
// [r * T, (r + 1) * T), runs interleaved granule by granule (physical p = t * runs + r), so that `runs` consumers
// that each walk one run read neighbouring addresses at the same time.  runs = 1: the sorted order itself.
// Per physical granule: psrc / pvalid (what the gather kernel copies), ppre (the granule word: the bucket's bin
// offset; with pack_rows -- ordered tables, histogram in LDS, offset < 2^24 -- the row count - 1 in the top byte),
// pkp = {bucket key, granule word} for the sparse kernel.
struct BucketedLayout {
  std::vector<unsigned> psrc, pvalid, ppre, pkp;
  size_t P = 0, A = 1;   // physical granules; array length (>= 1)
};

inline void bucketed_layout(const std::vector<unsigned>& lsrc, const std::vector<unsigned>& lvalid,
                            const std::vector<unsigned>& lwhich, const std::vector<unsigned>& keys,
                            const std::vector<unsigned>& key_pre, unsigned outside, int runs, bool pack_rows,
                            BucketedLayout& o) {
  const size_t L = lsrc.size();
  const size_t T = (L + (size_t)runs - 1) / (size_t)runs;
  o.P = T * (size_t)runs;
  o.A = std::max<size_t>(o.P, 1);
  o.psrc.assign(o.A, 0u);
  o.pvalid.assign(o.A, 0u);
  o.ppre.assign(o.A, 0u);
  o.pkp.assign(2 * o.A, 0u);
  for (size_t p = 0; p < o.P; p++) {
    const size_t r = p % (size_t)runs, t = p / (size_t)runs, l = r * T + t;
    if (l < L) {
      o.psrc[p] = lsrc[l];
      o.pvalid[p] = lvalid[l];
      o.ppre[p] = key_pre[lwhich[l]];
      if (pack_rows) o.ppre[p] |= (lvalid[l] - 1u) << 24;
      o.pkp[2 * p] = keys[lwhich[l]];
    } else {  // padding granule at the end of the last runs: no samples, stays in the last bucket
      o.pkp[2 * p] = L ? keys[lwhich[L - 1]] : outside;
    }
    o.pkp[2 * p + 1] = o.ppre[p];
  }
}

// The evaluator's event bins grouped by the buckets of a sort (fill_sparse_kernel): per bucket key an
// open-addressing table keyed by the event bin's index contribution of the WRITTEN observables (flat index minus
// the bucket's offset, canonical decomposition), value = the event bin's counter slot (its rank among the sorted
// distinct event bins).  dir[2 key] = the table's offset in tkeys / tslot, dir[2 key + 1] = log2(size) | probes << 8,
// or SXMC_SPARSE_EMPTY (no event bin in the bucket) / SXMC_SPARSE_SLOW (every sample through the global table).
struct BucketTables {
  std::vector<unsigned> dir, tkeys, tslot;
};

inline void bucket_tables(unsigned nkeys, unsigned mask, const unsigned* radix, const int* nbins, const int* stride,
                          int D, const std::vector<unsigned>& targets, BucketTables& o) {
  o.dir.assign(2 * ((size_t)nkeys + 1), 0u);
  o.tkeys.clear();
  o.tslot.clear();
  for (size_t k = 0; k <= nkeys; k++) o.dir[2 * k + 1] = SXMC_SPARSE_EMPTY;
  // bucket keys with an index equal to nbins: their samples alias into other rows of the flat index
  for (unsigned key = 0; key < nkeys; key++) {
    for (int k = 0; k < D; k++) {
      if (!((mask >> k) & 1u)) continue;
      if ((key / radix[k]) % ((unsigned)nbins[k] + 1u) == (unsigned)nbins[k]) o.dir[2 * (size_t)key + 1] = SXMC_SPARSE_SLOW;
    }
  }
  std::vector<std::vector<std::pair<unsigned, unsigned>>> by_key;   // only for keys that have some
  std::vector<int> where((size_t)nkeys, -1);
  for (size_t t = 0; t < targets.size(); t++) {
    const unsigned flat = targets[t];
    unsigned key = 0, pre = 0;
    for (int k = 0; k < D; k++) {
      if (!((mask >> k) & 1u)) continue;
      const unsigned idx = (flat / (unsigned)stride[k]) % (unsigned)nbins[k];
      key += idx * radix[k];
      pre += idx * (unsigned)stride[k];
    }
    if (where[key] < 0) {
      where[key] = (int)by_key.size();
      by_key.emplace_back();
    }
    by_key[(size_t)where[key]].push_back({flat - pre, (unsigned)t});
  }
  for (unsigned key = 0; key < nkeys; key++) {
    if (where[key] < 0) continue;
    const auto& list = by_key[(size_t)where[key]];
    // cells of four keys (one 16-byte LDS read per probe); load <= 25 %, or <= 50 % for the largest buckets
    int lg = std::max(2, ceil_log2(4 * list.size()));
    if (lg > SXMC_SPARSE_SMAX_LOG2) lg = std::max(2, ceil_log2(2 * list.size()));
    if (lg > SXMC_SPARSE_SMAX_LOG2) {                                  // more event bins than a wave's slice holds
      o.dir[2 * (size_t)key + 1] = SXMC_SPARSE_SLOW;
      continue;
    }
    const size_t off = o.tkeys.size(), S = (size_t)1 << lg, cells = S / 4;
    o.tkeys.resize(off + S, 0xFFFFFFFFu);
    o.tslot.resize(off + S, 0u);
    unsigned probes = 1;
    for (const auto& e : list) {
      size_t cell = lg > 2 ? (size_t)((e.first * 0x9E3779B1u) >> (34 - lg)) : 0;
      unsigned dist = 1;
      for (;; cell = (cell + 1) & (cells - 1), dist++) {
        size_t at = off + 4 * cell, free_slot = 4;
        for (size_t m = 0; m < 4; m++)
          if (o.tkeys[at + m] == 0xFFFFFFFFu) {
            free_slot = m;
            break;
          }
        if (free_slot < 4) {
          o.tkeys[at + free_slot] = e.first;
          o.tslot[at + free_slot] = e.second;
          break;
        }
      }
      probes = std::max(probes, dist);
    }
    if (probes > 255) {   // (cannot happen below 100 % load; keeps the field in range)
      o.dir[2 * (size_t)key + 1] = SXMC_SPARSE_SLOW;
      continue;
    }
    lg |= (int)(probes << 8);
    o.dir[2 * (size_t)key] = (unsigned)off;
    o.dir[2 * (size_t)key + 1] = (unsigned)lg;
  }
}

// ---------------------------------------------------------------------------------------------- event classes
// Events with the same bin (or counter slot) in every member contribute the same term to the event sum
// (nll_kernels.cpp:101-112), so the sum runs over the K distinct tuples, each weighted by its multiplicity.
// arr[j]: member j's table, one entry per event (E entries each).  tables[j * K + k] = member j's entry of class k.
struct EventClasses {
  size_t K = 0;
  std::vector<unsigned> first, weight;   // per class: an event that belongs to it; how many do
  std::vector<int> tables;
};

inline void event_classes(const std::vector<const std::vector<int>*>& arr, size_t E, EventClasses& o) {
  const size_t S = arr.size();
  // members with identical tables (one binning, one data set: the usual case) count once in the key
  std::vector<int> distinct;
  for (size_t j = 0; j < S; j++) {
    bool seen = false;
    for (size_t q = 0; q < distinct.size() && !seen; q++) seen = *arr[(size_t)distinct[q]] == *arr[j];
    if (!seen) distinct.push_back((int)j);
  }
  std::vector<unsigned> order(E);
  for (size_t i = 0; i < E; i++) order[i] = (unsigned)i;
  auto less = [&](unsigned a, unsigned b) {
    for (int q : distinct) {
      const int x = (*arr[(size_t)q])[a], y = (*arr[(size_t)q])[b];
      if (x != y) return x < y;
    }
    return false;
  };
  auto same = [&](unsigned a, unsigned b) {
    for (int q : distinct)
      if ((*arr[(size_t)q])[a] != (*arr[(size_t)q])[b]) return false;
    return true;
  };
  std::sort(order.begin(), order.end(), less);
  o.first.clear();
  o.weight.clear();
  for (size_t i = 0; i < E; i++) {
    if (i > 0 && same(order[i - 1], order[i])) {
      o.weight.back()++;
    } else {
      o.first.push_back(order[i]);
      o.weight.push_back(1u);
    }
  }
  o.K = o.first.size();
  o.tables.assign(std::max<size_t>(S * o.K, 1), 0);
  for (size_t j = 0; j < S; j++)
    for (size_t k = 0; k < o.K; k++) o.tables[j * o.K + k] = (*arr[j])[o.first[k]];
}

// CODES (fill_ordered_body): the window [base, base + 65532 step] of every streamed field of an ordered table, from the
// fields' finite ranges in the table (minmax[2m], minmax[2m + 1]; min > max: no finite value) and the domains of the
// observables among them (fields 0 .. nobs-1).  An observable's window is its finite range cut to its domain widened by
// its own width on either side -- a value further out needs a scale or shift of the order of the whole domain to come
// back in, and its row is marked "ask the exact columns" instead; a field that is only read gets its finite range, cut
// to three widths of the observables' windows around them when the two overlap at all.  step > 0 and finite, always.
struct CodeWindows {
  std::vector<double> base, step;
};
inline void code_windows(const float* minmax, int nfields, int nobs, const double* lower, const double* upper,
                         CodeWindows& o) {
  o.base.assign((size_t)nfields, 0.0);
  o.step.assign((size_t)nfields, 1.0);
  double ulo = 0, uhi = -1;   // union of the observables' windows
  for (int m = 0; m < nfields; m++) {
    double wlo = minmax[2 * m], whi = minmax[2 * m + 1];
    const bool none = !(wlo <= whi);
    if (m < nobs) {
      const double lo = lower[m], hi = upper[m], w = hi - lo;
      wlo = none ? lo - w : std::max(wlo, lo - w);
      whi = none ? hi + w : std::min(whi, hi + w);
      if (!(wlo < whi)) {   // (no finite value near the domain)
        wlo = lo - w;
        whi = hi + w;
      }
      if (uhi < ulo) {
        ulo = wlo;
        uhi = whi;
      } else {
        ulo = std::min(ulo, wlo);
        uhi = std::max(uhi, whi);
      }
    } else if (none) {
      wlo = 0;
      whi = 1;
    } else if (ulo <= uhi) {
      const double w = uhi - ulo, clo = std::max(wlo, ulo - 3 * w), chi = std::min(whi, uhi + 3 * w);
      if (clo < chi) {
        wlo = clo;
        whi = chi;
      }
    }
    double step = (whi - wlo) / 65532.0;   // (the largest value lands in code 65532 of 0 .. 65533)
    if (!(step > 0) || !std::isfinite(step)) step = std::max(std::fabs(wlo), 1.0) * 0x1p-20;
    o.base[(size_t)m] = wlo;
    o.step[(size_t)m] = step;
  }
}

// LDS of fill_ordered_body (histograms in LDS): 4 header words, per chain 2^rlog replicas `rstride` words apart, 64
// spare words, then the queues of the codes path: 4 words + 2 per entry, (1 << qlog) entries shared out over the waves.
inline unsigned ordered_rstride_plain(int max_bins) { return (((unsigned)max_bins + 63u) & ~63u) + 16u; }
// ... in the padded form the codes path uses when ONE observable is binned per sample and it is the histogram's
// outermost dimension: (nbins + 2) rows of S' words, S' = S | 1 (odd: lanes that differ in the index hit different
// banks), a guard row either side; + 16 mod 64 like the plain form (a bin's replicas in different banks)
inline unsigned ordered_rstride_padded(int total_bins, int outer_bins) {
  const unsigned S = (unsigned)(total_bins / outer_bins), Sp = S | 1u;
  const unsigned words = ((unsigned)outer_bins + 2u) * Sp;
  return ((words + 63u) & ~63u) + 16u;
}
inline size_t ordered_queue_bytes(unsigned qlog) { return qlog ? (4 + ((size_t)2 << qlog)) * 4 : 0; }

}  // namespace sxplan
