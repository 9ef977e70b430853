// test_plan.cpp -- device-free tests of the host planners of libsxmc_hip.so (sxmc_amd/csrc/sxmc_plan.h): every
// table the gfx950 kernels index blindly, built from randomized shapes -- ragged and empty members, no events, all
// events outside the domain or of another data set, bucket offsets at the 24-bit limit of the granule word -- and
// then walked exactly as the kernels walk them (same hashes, same probe sequences, same loop bounds), checking that
// nothing is missed, counted twice or addressed out of range.  Plain g++; run by tests/test_plan_cpu.py in a plain
// build and under AddressSanitizer + UndefinedBehaviorSanitizer.  No HIP header, no device.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>

#include "../../sxmc_amd/csrc/sxmc_plan.h"
#include "mini_test.h"

namespace {
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  uint64_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  }
  uint64_t below(uint64_t n) { return n ? next() % n : 0; }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// what fill_body / fill_ordered_body / fill_sparse_body do with a workgroup's segment list: lane tid starts at
// v0 + tid and strides by `step` while below v1; every lane runs ceil((v1 - v0) / step) stages and clamps the index
// of a load past the end to v1 - 1
void replay_partition(const std::vector<unsigned long long>& nvec, int grid, int threads,
                      const std::vector<SxSegment>& segs, const std::vector<unsigned>& blk_off,
                      unsigned long long align) {
  EXPECT_EQ((size_t)grid + 1, blk_off.size());
  EXPECT_EQ(0u, blk_off[0]);
  EXPECT_EQ((unsigned)segs.size(), blk_off.back());
  std::vector<std::vector<unsigned char>> seen(nvec.size());
  for (size_t j = 0; j < nvec.size(); j++) seen[j].assign((size_t)nvec[j], 0);
  for (int wg = 0; wg < grid; wg++) {
    EXPECT_TRUE(blk_off[(size_t)wg] <= blk_off[(size_t)wg + 1]);
    for (unsigned si = blk_off[(size_t)wg]; si < blk_off[(size_t)wg + 1]; si++) {
      const SxSegment& sg = segs[si];
      EXPECT_TRUE(sg.sig >= 0 && (size_t)sg.sig < nvec.size());
      EXPECT_TRUE(sg.v0 < sg.v1);                       // vlast = v1 - 1 is a valid unit: the clamped loads stay inside
      EXPECT_TRUE(sg.v1 <= nvec[(size_t)sg.sig]);
      EXPECT_TRUE(sg.step >= (unsigned long long)threads && sg.step % (unsigned long long)threads == 0);
      EXPECT_EQ(0ull, sg.v0 % align);
      for (int tid = 0; tid < threads; tid++) {
        for (unsigned long long v = sg.v0 + (unsigned long long)tid; v < sg.v1; v += sg.step) {
          seen[(size_t)sg.sig][(size_t)v]++;
        }
      }
    }
  }
  for (size_t j = 0; j < nvec.size(); j++)
    for (unsigned char c : seen[j]) EXPECT_EQ(1, (int)c);   // every unit exactly once
}
}  // namespace

TEST(Plan, PartitionsCoverEveryUnitExactlyOnce) {
  Rng r(1);
  for (int trial = 0; trial < 400; trial++) {
    const int nmembers = (int)r.below(13);
    const unsigned long long align = r.below(3) == 0 ? 64 : 1;
    std::vector<unsigned long long> nvec;
    for (int j = 0; j < nmembers; j++) {
      unsigned long long n = r.below(4) == 0 ? 0 : r.below(r.below(2) ? 40000 : 300);
      if (align > 1) n = n / align * align;
      nvec.push_back(n);
    }
    const int threads = 64 * (1 + (int)r.below(16));
    const int grid = 1 + (int)r.below(r.below(2) ? 600 : 12);
    for (int mode = 0; mode <= 2; mode++) {
      std::vector<SxSegment> segs;
      std::vector<unsigned> blk_off;
      int built = 0;
      sxplan::build_partition(nvec, grid, threads, mode, segs, blk_off, built, align);
      EXPECT_TRUE(built == 1 || built == 2);
      if (mode == 1) EXPECT_EQ(1, built);
      // (interleaved chunks start on multiples of `threads`: alignment to granules holds when threads % 64 == 0)
      replay_partition(nvec, grid, threads, segs, blk_off, built == 2 ? std::min<unsigned long long>(align, 64) : align);
      if (built == 2) {
        // every workgroup reads ONE member: one histogram flush per workgroup
        for (int wg = 0; wg < grid; wg++) EXPECT_TRUE(blk_off[(size_t)wg + 1] - blk_off[(size_t)wg] <= 1u);
      }
      // teams of workgroups over contiguous parts of a member (tables sorted by bin)
      for (int groups : {2, 3, 7, 50}) {
        sxplan::build_partition(nvec, grid, threads, mode, segs, blk_off, built, align, groups);
        replay_partition(nvec, grid, threads, segs, blk_off, built == 2 ? std::min<unsigned long long>(align, 64) : align);
        if (built == 2) {
          for (int wg = 0; wg < grid; wg++) EXPECT_TRUE(blk_off[(size_t)wg + 1] - blk_off[(size_t)wg] <= 1u);
        }
      }
    }
  }
}

TEST(Plan, ApportionmentGivesEveryNonEmptyMemberAWorkgroupAndNeverMoreThanItsChunks) {
  Rng r(2);
  for (int trial = 0; trial < 2000; trial++) {
    std::vector<unsigned long long> sizes;
    const int n = 1 + (int)r.below(24);
    for (int j = 0; j < n; j++) sizes.push_back(r.below(3) == 0 ? 0 : 1 + r.below(1000000));
    const int threads = 64 * (1 + (int)r.below(16)), grid = 1 + (int)r.below(1024);
    std::vector<int> K;
    const bool fits = sxplan::apportion_workgroups(sizes, grid, threads, K);
    int used = 0, nonempty = 0;
    for (int j = 0; j < n; j++) {
      used += K[(size_t)j];
      if (sizes[(size_t)j]) {
        nonempty++;
        EXPECT_TRUE(K[(size_t)j] >= 1);
        EXPECT_TRUE((unsigned long long)K[(size_t)j] <= (sizes[(size_t)j] + threads - 1) / threads);
      } else {
        EXPECT_EQ(0, K[(size_t)j]);
      }
    }
    EXPECT_EQ(fits, used <= grid);
    if (grid >= 2 * nonempty) EXPECT_TRUE(fits);
  }
}

TEST(Plan, EvalPointBinsFollowTheReferenceLoop) {
  // pdfz.cpp:264-301 on a 3-D geometry with uneven bins: interior points, both edges, NaN, another data set, and an
  // index that rounds up to one past the end
  const double lower[3] = {0.0, -1.0, 5.0}, upper[3] = {10.0, 1.0, 15.0};
  const int nbins[3] = {7, 3, 11}, stride[3] = {33, 11, 1};
  double scale[3];
  for (int k = 0; k < 3; k++) scale[k] = (double)nbins[k] / (upper[k] - lower[k]);
  Rng r(3);
  std::vector<float> pts;
  std::vector<int> want;
  for (int i = 0; i < 5000; i++) {
    float x[3];
    bool in = true;
    int bin = 0;
    for (int k = 0; k < 3; k++) {
      const int kind = (int)r.below(12);
      double v = lower[k] + (upper[k] - lower[k]) * r.uni();
      if (kind == 0) v = lower[k];
      if (kind == 1) v = upper[k];
      if (kind == 2) v = lower[k] - 0.001;
      if (kind == 3) v = std::nan("");
      if (kind == 4) v = std::nextafter((float)upper[k], 0.0f);
      x[k] = (float)v;
      const double e = (double)x[k];
      if (!(e >= lower[k] && e < upper[k])) in = false;
      if (in) bin += (int)((e - lower[k]) * scale[k]) * stride[k];
    }
    const float ds = r.below(5) == 0 ? 3.0f : 2.0f;
    for (int k = 0; k < 3; k++) pts.push_back(x[k]);
    pts.push_back(ds);
    if (in && bin >= 7 * 3 * 11) in = false;
    want.push_back(!in ? -1 : ds != 2.0f ? -2 : bin);
  }
  std::vector<int> rb;
  sxplan::eval_point_bins(pts.data(), want.size(), 3, lower, upper, scale, stride, 7 * 3 * 11, 2u, rb);
  EXPECT_EQ(want.size(), rb.size());
  for (size_t i = 0; i < want.size(); i++) EXPECT_EQ(want[i], rb[i]);
  // the precedence the reference has: outside the domain wins over "another data set" (pdfz.cpp:289-300)
  const float outside_other[4] = {-5.0f, 0.0f, 6.0f, 9.0f};
  sxplan::eval_point_bins(outside_other, 1, 3, lower, upper, scale, stride, 231, 2u, rb);
  EXPECT_EQ(-1, rb[0]);
  sxplan::eval_point_bins(nullptr, 0, 3, lower, upper, scale, stride, 231, 2u, rb);   // no events
  EXPECT_EQ((size_t)0, rb.size());
}

namespace {
// fill_kernels.inc.h: sparse_count + sparse_lookup
int device_lookup(const sxplan::SparseTables& t, unsigned bin) {
  const unsigned hc = (bin * 0xC2B2AE35u) >> (32 - t.cbits), hd = (bin * 0x27D4EB2Fu) >> (32 - t.cbits);
  if (!((t.coarse.at(hc >> 5) >> (hc & 31u)) & (t.coarse.at(hd >> 5) >> (hd & 31u)) & 1u)) return -1;
  const unsigned hb = (bin * 0x9E3779B1u) >> (32 - t.fbits);
  if (!((t.filter.at(hb >> 5) >> (hb & 31u)) & 1u)) return -1;
  const unsigned mask = (1u << t.tbits) - 1u;
  unsigned hp = (bin * 0x85EBCA6Bu) >> (32 - t.tbits);
  for (unsigned probe = 0; probe <= mask; probe++) {
    const unsigned k = t.table.at(2 * (size_t)hp), v = t.table.at(2 * (size_t)hp + 1);
    if (k == bin) return (int)v;
    if (k == 0xFFFFFFFFu) return -1;
    hp = (hp + 1u) & mask;
  }
  return -2;   // the probe loop ran the whole table: must not happen (load <= 50 %)
}
}  // namespace

TEST(Plan, SparseTablesFindEveryEventBinAndNothingElse) {
  Rng r(4);
  for (int trial = 0; trial < 60; trial++) {
    const size_t E = trial == 0 ? 0 : (size_t)r.below(trial % 7 == 0 ? 200000 : 3000);
    const unsigned nbins = 1u + (unsigned)r.below(trial % 3 ? 1u << 27 : 5000);
    std::vector<int> rb(E);
    for (size_t i = 0; i < E; i++) {
      const int kind = trial == 1 ? 0 : trial == 2 ? 1 : (int)r.below(10);   // all -1 / all -2 / mixed
      rb[i] = kind == 0 ? -1 : kind == 1 ? -2 : (int)r.below(nbins);
    }
    sxplan::SparseTables t;
    sxplan::build_sparse_tables(rb, t);
    std::set<unsigned> distinct;
    for (int b : rb)
      if (b >= 0) distinct.insert((unsigned)b);
    EXPECT_EQ(distinct.size(), t.targets.size());
    EXPECT_EQ((size_t)2 << t.tbits, t.table.size());
    EXPECT_TRUE(((size_t)1 << t.tbits) >= 2 * t.targets.size());           // load <= 50 %: probing terminates
    EXPECT_TRUE(t.cbits <= 20 && t.cbits >= 10 && t.fbits <= 26 && t.fbits >= 16);
    for (size_t i = 0; i < E; i++) {
      if (rb[i] < 0) {
        EXPECT_EQ(rb[i], t.slot[i]);
      } else {
        EXPECT_TRUE(t.slot[i] >= 0 && (size_t)t.slot[i] < t.targets.size());
        EXPECT_EQ((unsigned)rb[i], t.targets[(size_t)t.slot[i]]);
        EXPECT_EQ(t.slot[i], device_lookup(t, (unsigned)rb[i]));
      }
    }
    for (int q = 0; q < 3000; q++) {        // bins that hold no event: never a counter, never an endless probe
      const unsigned b = (unsigned)r.below(nbins);
      if (distinct.count(b)) continue;
      EXPECT_EQ(-1, device_lookup(t, b));
    }
  }
}

TEST(Plan, BucketGranulesAndLayoutsPlaceEveryKeptRowOnce) {
  Rng r(5);
  for (int trial = 0; trial < 120; trial++) {
    const unsigned nkeys = 1u + (unsigned)r.below(trial % 2 ? 400 : 12);
    const size_t n = trial == 0 ? 1 : (size_t)r.below(trial % 5 == 0 ? 300000 : 20000) + 1;
    // rows sorted by key; key == nkeys ("outside") sorts last
    std::vector<unsigned> keys(n);
    const bool has_outside = r.below(2) != 0, sparse_keys = r.below(3) == 0;
    for (size_t i = 0; i < n; i++) {
      unsigned k = (unsigned)r.below(nkeys + (has_outside ? 1 : 0));
      if (sparse_keys && k < nkeys) k = k / 7 * 7;
      keys[i] = k;
    }
    std::sort(keys.begin(), keys.end());
    std::vector<unsigned> first((size_t)nkeys + 1, 0xFFFFFFFFu);
    for (size_t i = n; i-- > 0;) first[keys[i]] = (unsigned)i;   // (sx_bucket_first: first row of every key)
    sxplan::GranulePlan g;
    sxplan::bucket_granules(first, nkeys, n, g);
    const size_t kept = (size_t)(std::lower_bound(keys.begin(), keys.end(), nkeys) - keys.begin());
    EXPECT_EQ(kept, g.kept);
    if (!g.worth_it) {
      EXPECT_TRUE(g.lsrc.empty());
      continue;
    }
    std::vector<unsigned char> seen(kept, 0);
    for (size_t l = 0; l < g.lsrc.size(); l++) {
      EXPECT_TRUE(g.lvalid[l] >= 1 && g.lvalid[l] <= 256);
      EXPECT_TRUE(g.lwhich[l] < g.present.size());
      if (l) EXPECT_TRUE(g.lwhich[l] >= g.lwhich[l - 1]);
      for (unsigned q = 0; q < g.lvalid[l]; q++) {
        const size_t row = (size_t)g.lsrc[l] + q;
        EXPECT_TRUE(row < kept);
        EXPECT_EQ(g.present[g.lwhich[l]], keys[row]);     // a granule never straddles two buckets
        seen[row]++;
      }
    }
    for (unsigned char c : seen) EXPECT_EQ(1, (int)c);
    // bucket offsets: two untouched observables of a 3-D problem (mask 0b101), keys mixed radix with bases nbins + 1
    const int nb[3] = {(int)std::max(1u, nkeys / 5), 9, 4}, stride[3] = {36, 4, 1};
    unsigned radix[SXMC_MAX_NFIELDS] = {0};
    radix[2] = 1;
    radix[0] = (unsigned)nb[2] + 1;
    std::vector<unsigned> key_pre;
    sxplan::bucket_key_offsets(g.present, 0x5u, radix, nb, stride, 3, key_pre);
    for (size_t i = 0; i < g.present.size(); i++) {
      const unsigned i2 = g.present[i] % ((unsigned)nb[2] + 1), i0 = (g.present[i] / radix[0]) % ((unsigned)nb[0] + 1);
      EXPECT_EQ(i0 * 36u + i2, key_pre[i]);
    }
    // the physical order for several run counts: a bijection onto the logical granules + empty padding granules
    for (int runs : {1, 3, 8, 64, 1000}) {
      for (int pack = 0; pack < 2; pack++) {
        std::vector<unsigned> pre = key_pre;
        if (pack) {
          for (unsigned& p : pre) p &= 0xFFFFFFu;
          if (!pre.empty()) pre[0] = 0xFFFFFFu;            // the largest offset the granule word can carry
        }
        sxplan::BucketedLayout lay;
        sxplan::bucketed_layout(g.lsrc, g.lvalid, g.lwhich, g.present, pre, nkeys, runs, pack != 0, lay);
        EXPECT_EQ(0u, (unsigned)(lay.P % (size_t)runs));
        EXPECT_TRUE(lay.P >= g.lsrc.size() && lay.P < g.lsrc.size() + (size_t)runs);
        EXPECT_EQ(lay.A, lay.psrc.size());
        EXPECT_EQ(2 * lay.A, lay.pkp.size());
        std::multiset<unsigned> srcs;
        for (size_t p = 0; p < lay.P; p++) {
          if (lay.pvalid[p] == 0) {                        // padding: no rows, a key that exists (or `outside`)
            EXPECT_TRUE(lay.pkp[2 * p] == nkeys || std::binary_search(g.present.begin(), g.present.end(), lay.pkp[2 * p]));
            continue;
          }
          srcs.insert(lay.psrc[p]);
          const size_t T = lay.P / (size_t)runs, l = (p % (size_t)runs) * T + p / (size_t)runs;
          EXPECT_EQ(g.lsrc[l], lay.psrc[p]);
          EXPECT_EQ(g.present[g.lwhich[l]], lay.pkp[2 * p]);
          EXPECT_EQ(lay.ppre[p], lay.pkp[2 * p + 1]);
          if (pack) {                                      // fill_ordered_body: rows - 1 in the top byte
            EXPECT_EQ(lay.pvalid[p], (lay.ppre[p] >> 24) + 1u);
            EXPECT_EQ(pre[g.lwhich[l]], lay.ppre[p] & 0xFFFFFFu);
          } else {
            EXPECT_EQ(pre[g.lwhich[l]], lay.ppre[p]);
          }
        }
        EXPECT_EQ(g.lsrc.size(), srcs.size());
      }
    }
  }
  // a table that is mostly padding (many buckets, few rows) is refused
  std::vector<unsigned> first(5001, 0xFFFFFFFFu);
  for (unsigned k = 0; k < 5000; k++) first[k] = k;
  sxplan::GranulePlan g;
  sxplan::bucket_granules(first, 5000, 5000, g);
  EXPECT_TRUE(!g.worth_it);
}

TEST(Plan, BucketTablesAreFoundByTheKernelsProbeSequence) {
  Rng r(6);
  for (int trial = 0; trial < 80; trial++) {
    // D observables; `mask` = the untouched ones (the bucket key), the others are written and binned per sample
    const int D = 2 + (int)r.below(4);
    int nbins[SXMC_MAX_NFIELDS] = {0}, stride[SXMC_MAX_NFIELDS] = {0};
    unsigned radix[SXMC_MAX_NFIELDS] = {0};
    unsigned mask = 0;
    for (int k = 0; k < D; k++) {
      nbins[k] = 1 + (int)r.below(trial % 4 == 0 ? 40 : 9);
      if (r.below(2)) mask |= 1u << k;
    }
    if (mask == 0) mask = 1u;
    if (mask == (1u << D) - 1u) mask &= ~2u;
    long long total = 1;
    for (int k = D - 1; k >= 0; k--) {
      stride[k] = (int)total;
      total *= nbins[k];
    }
    unsigned long long nkeys = 1;
    for (int k = D - 1; k >= 0; k--) {
      if (!((mask >> k) & 1u)) continue;
      radix[k] = (unsigned)nkeys;
      nkeys *= (unsigned long long)nbins[k] + 1ull;
    }
    // distinct event bins; some trials crowd one bucket beyond a wave's slice (> 256 event bins in it)
    std::set<unsigned> tset;
    const size_t want = (size_t)r.below((size_t)std::min<long long>(total, trial % 5 == 0 ? 4000 : 300)) + (trial ? 1 : 0);
    while (tset.size() < want && tset.size() < (size_t)total) tset.insert((unsigned)r.below((uint64_t)total));
    std::vector<unsigned> targets(tset.begin(), tset.end());
    sxplan::BucketTables bt;
    sxplan::bucket_tables((unsigned)nkeys, mask, radix, nbins, stride, D, targets, bt);
    EXPECT_EQ(2 * ((size_t)nkeys + 1), bt.dir.size());
    EXPECT_EQ(bt.tkeys.size(), bt.tslot.size());
    std::set<unsigned> keys_with_events;
    for (size_t t = 0; t < targets.size(); t++) {
      unsigned key = 0, pre = 0;
      for (int k = 0; k < D; k++) {
        if (!((mask >> k) & 1u)) continue;
        const unsigned idx = (targets[t] / (unsigned)stride[k]) % (unsigned)nbins[k];
        key += idx * radix[k];
        pre += idx * (unsigned)stride[k];
      }
      keys_with_events.insert(key);
      const unsigned off = bt.dir[2 * (size_t)key], info = bt.dir[2 * (size_t)key + 1];
      const unsigned lg = info & 0xFFu, probes = (info >> 8) & 0xFFu;
      if (lg == SXMC_SPARSE_SLOW) continue;                      // every sample of this bucket: the global table
      EXPECT_TRUE(lg >= 2 && lg <= SXMC_SPARSE_SMAX_LOG2 && probes >= 1);
      const unsigned p2 = targets[t] - pre;                       // the written observables' contribution
      const unsigned cshift = 34u - lg, cmask = (1u << (lg - 2u)) - 1u;
      const unsigned cell = lg > 2u ? (p2 * 0x9E3779B1u) >> cshift : 0u;
      int found = -1;
      for (unsigned pr = 0; pr < probes && found < 0; pr++) {
        for (unsigned m = 0; m < 4; m++) {
          const size_t at = (size_t)off + 4u * ((cell + pr) & cmask) + m;
          EXPECT_TRUE(at < bt.tkeys.size());
          if (bt.tkeys[at] == p2) found = (int)bt.tslot[at];
        }
      }
      EXPECT_EQ((int)t, found);                                   // slot = rank among the sorted distinct event bins
    }
    for (unsigned key = 0; key < (unsigned)nkeys; key++) {
      bool has_edge_index = false;
      for (int k = 0; k < D; k++) {
        if (!((mask >> k) & 1u)) continue;
        if ((key / radix[k]) % ((unsigned)nbins[k] + 1u) == (unsigned)nbins[k]) has_edge_index = true;
      }
      const unsigned info = bt.dir[2 * (size_t)key + 1] & 0xFFu;
      if (has_edge_index) {
        EXPECT_EQ((unsigned)SXMC_SPARSE_SLOW, info);
      } else if (!keys_with_events.count(key)) {
        EXPECT_EQ((unsigned)SXMC_SPARSE_EMPTY, info);
      } else {
        EXPECT_TRUE(info == SXMC_SPARSE_SLOW || info <= SXMC_SPARSE_SMAX_LOG2);
      }
    }
  }
}

TEST(Plan, EventClassesKeepTheSumOverEvents) {
  Rng r(7);
  for (int trial = 0; trial < 60; trial++) {
    const size_t S = 1 + (size_t)r.below(12), E = trial == 0 ? 0 : (size_t)r.below(5000);
    const int nbins = 1 + (int)r.below(trial % 2 ? 30 : 100000);
    std::vector<std::vector<int>> tab(S, std::vector<int>(E));
    for (size_t j = 0; j < S; j++) {
      if (j > 0 && r.below(2)) {
        tab[j] = tab[(size_t)r.below(j)];                 // members with identical tables (one binning, one data set)
        continue;
      }
      for (size_t i = 0; i < E; i++) tab[j][i] = r.below(8) == 0 ? -(int)(1 + r.below(2)) : (int)r.below((uint64_t)nbins);
    }
    std::vector<const std::vector<int>*> arr;
    for (const std::vector<int>& t : tab) arr.push_back(&t);
    sxplan::EventClasses c;
    sxplan::event_classes(arr, E, c);
    EXPECT_EQ(c.K, c.weight.size());
    EXPECT_TRUE(c.tables.size() >= S * c.K);
    // the multiset of per-event tuples is the multiset of class tuples repeated `weight` times
    std::map<std::vector<int>, unsigned> by_event, by_class;
    for (size_t i = 0; i < E; i++) {
      std::vector<int> tup;
      for (size_t j = 0; j < S; j++) tup.push_back(tab[j][i]);
      by_event[tup]++;
    }
    size_t total = 0;
    for (size_t k = 0; k < c.K; k++) {
      std::vector<int> tup;
      for (size_t j = 0; j < S; j++) tup.push_back(c.tables[j * c.K + k]);
      EXPECT_EQ(0u, by_class[tup]);                       // classes are distinct
      by_class[tup] = c.weight[k];
      total += c.weight[k];
    }
    EXPECT_EQ(E, total);
    EXPECT_TRUE(by_event == by_class);
  }
}

// CODES: the windows the table's 16-bit codes are taken in, and the LDS arithmetic of the padded histogram -- what
// column_codes_kernel and fill_ordered_body assume of them.
TEST(Plan, CodeWindowsHoldEveryValueTheyClaimAndTheirStepsArePositive) {
  Rng r(11);
  for (int trial = 0; trial < 400; trial++) {
    const int nfields = 2 + (int)r.below(3), nobs = 1 + (int)r.below((uint64_t)nfields - 1);
    std::vector<double> lo((size_t)nobs), hi((size_t)nobs);
    std::vector<float> mm(2 * (size_t)nfields);
    for (int k = 0; k < nobs; k++) {
      lo[(size_t)k] = (r.uni() - 0.5) * (r.below(3) ? 20.0 : 2e6);
      hi[(size_t)k] = lo[(size_t)k] + (r.below(4) ? 1.0 + 9.0 * r.uni() : 1e-3);
    }
    for (int m = 0; m < nfields; m++) {
      const int kind = (int)r.below(6);
      const double c = m < nobs ? lo[(size_t)m] : lo[0], w = m < nobs ? hi[(size_t)m] - lo[(size_t)m] : hi[0] - lo[0];
      double a = c - w * r.uni(), b = c + w * (1 + r.uni());
      if (kind == 0) { a = 1; b = -1; }                       // no finite value at all
      if (kind == 1) { a = b = c + 0.3 * w; }                 // one value
      if (kind == 2) { a = -3e38; b = 3e38; }                 // outliers at the ends of the float range
      if (kind == 3) { a = c + 50 * w; b = c + 60 * w; }      // everything far from the domain
      mm[2 * (size_t)m] = (float)a;
      mm[2 * (size_t)m + 1] = (float)b;
    }
    sxplan::CodeWindows cw;
    sxplan::code_windows(mm.data(), nfields, nobs, lo.data(), hi.data(), cw);
    EXPECT_EQ((size_t)nfields, cw.base.size());
    for (int m = 0; m < nfields; m++) {
      const double base = cw.base[(size_t)m], step = cw.step[(size_t)m];
      EXPECT_TRUE(step > 0 && std::isfinite(step) && std::isfinite(base));
      const double top = base + 65532.0 * step;
      EXPECT_TRUE(std::isfinite(top));
      // the kernel's bound on a field's magnitude inside the window is finite
      EXPECT_TRUE(std::isfinite(std::max(std::fabs(base), std::fabs(base + 65534.0 * step))));
      if (m < nobs) {
        // an observable's window never reaches further than one domain width beyond the domain ...
        const double w = hi[(size_t)m] - lo[(size_t)m];
        EXPECT_TRUE(base >= lo[(size_t)m] - w * (1 + 1e-12) && top <= hi[(size_t)m] + w * (1 + 1e-9));
        // ... and holds every finite value of the table inside that reach: code = floor((x - base) / step) <= 65533
        const double a = mm[2 * (size_t)m], b = mm[2 * (size_t)m + 1];
        if (a <= b) {
          const double x0 = std::max(a, lo[(size_t)m] - w), x1 = std::min(b, hi[(size_t)m] + w);
          if (x0 <= x1) {
            EXPECT_TRUE((x0 - base) / step >= 0.0 && (x1 - base) / step < 65534.0);
          }
        }
      }
    }
  }
}

TEST(Plan, PaddedHistogramRowsDoNotOverlapAndFitTheirReplica) {
  // fill_ordered_body, `outer`: bin idx * S + r lives in word (idx + 1) * S' + r, S' = S | 1; rows -1 and nbins are
  // guard rows; the flush walks words S' .. (nbins + 1) * S' - 1 and skips the pad word r = S of every row
  Rng r(13);
  for (int trial = 0; trial < 300; trial++) {
    const int nb = 1 + (int)r.below(200), S = 1 + (int)r.below(trial % 3 ? 500 : 40000 / nb);
    const int B = nb * S;
    if (B > 40832) continue;
    const unsigned Sp = (unsigned)S | 1u, rs = sxplan::ordered_rstride_padded(B, nb);
    EXPECT_TRUE(Sp >= (unsigned)S && (Sp & 1u) == 1u);
    EXPECT_TRUE(rs >= ((unsigned)nb + 2u) * Sp && rs % 64u == 16u);
    std::set<unsigned> words;
    for (int idx = -1; idx <= nb; idx++) {
      for (int rr = 0; rr < S; rr += 1 + (int)r.below(7)) {
        const unsigned w = (unsigned)(idx + 1) * Sp + (unsigned)rr;
        EXPECT_TRUE(w < rs);
        EXPECT_TRUE(words.insert(w).second);              // no two (idx, r) share a word
        if (idx >= 0 && idx < nb) {
          // the flush's way back: idx = (w - S') / S' with a single-precision reciprocal, r = the rest
          const unsigned wf = w - Sp;
          const unsigned q = (unsigned)(((float)wf + 0.5f) * (1.0f / (float)Sp));
          EXPECT_EQ((unsigned)idx, q);
          EXPECT_EQ((unsigned)rr, wf - q * Sp);
          // ... and the exact path's way there: word of bin b = b + S' + (S' - S) * (b / S)
          const unsigned b = (unsigned)idx * (unsigned)S + (unsigned)rr;
          const unsigned qb = (unsigned)(((float)b + 0.5f) * (1.0f / (float)S));
          EXPECT_EQ((unsigned)idx, qb);
          EXPECT_EQ(w, b + Sp + (Sp - (unsigned)S) * qb);
        }
      }
    }
    EXPECT_TRUE(sxplan::ordered_queue_bytes(9) == (4 + 1024) * 4 && sxplan::ordered_queue_bytes(0) == 0);
  }
}

int main(int argc, char** argv) { return mini::run_all(argc > 1 ? argv[1] : nullptr); }
