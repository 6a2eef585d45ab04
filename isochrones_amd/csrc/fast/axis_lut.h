// Bucket tables for the LDS-staged axes: where the bisection of an axis starts, and how wide it has to be.
// Host side only (the device side is lut_start() in brackets.h).
//
// The reference bisects every axis from scratch for every sample (interp.py:10-35: searchsorted over all n nodes,
// 8 levels for the 196 masses, 7 for the 70 effective temperatures).  A bracket is the integer
//     i(x) = #{ j : a_j <= x } - 1                                   (clamped to [0, n - 2])
// and any *monotone* integer function bucket(x) cuts the search down without changing that integer: if
// bucket(a_j) < bucket(x) then a_j < x, so  start(b) = #{ j : bucket(a_j) < b } - 1  is a node at or below every x
// of bucket b, and the only nodes that can still be <= x are the ones whose own bucket is b.  With
// F = max_b #{ j : bucket(a_j) = b } the bracket lies in the window [start, start + F], which the kernels bisect in
// ceil(log2(F + 1)) levels - whatever rounding the bucket function suffers, as long as host and device evaluate the
// same one.  The function used here costs three integer instructions:
//     bucket(x) = (hi32(x + c) >> sh) - b0          (c: a double whose low word is zero, so one scalar register)
// the exponent and leading mantissa bits of a positive double (x + c > 0): piecewise-logarithmic buckets for c = 0
// (masses 0.1 .. 300, temperatures 2500 .. 200000) and uniform ones when c lifts the axis into a single binade
// ([Fe/H], log g, log age); x -> x + c is monotone under IEEE rounding and so is everything after it.
// Tables are bytes (node indices < 256; an axis with more nodes keeps the full bisection: one bucket, window n).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace iso {

struct AxisLutChoice {
    double c;
    int sh, b0, nbk, win;             // win = F + 1 nodes in the window
    int levels;                       // ceil(log2(win))
};

inline int32_t lut_hi32(double v)
{
    int64_t bits;
    std::memcpy(&bits, &v, sizeof(bits));
    return (int32_t)(bits >> 32);
}

// c rounded up to a double whose low word is zero (the kernels get its high word only: one scalar register)
inline double lut_hi_only(double c)
{
    int64_t bits;
    std::memcpy(&bits, &c, sizeof(bits));
    if ((bits & 0xFFFFFFFFll) == 0) return c;
    bits &= ~0xFFFFFFFFll;
    if (c > 0.0) bits += 0x100000000ll;         // away from zero for c > 0, towards zero for c < 0: up either way
    double r;
    std::memcpy(&r, &bits, sizeof(r));
    return r;
}

inline int lut_bucket(double x, double c, int sh, int b0) { return (lut_hi32(x + c) >> sh) - b0; }

inline int lut_levels(int win)
{
    int lv = 0;
    while ((1 << lv) < win) ++lv;
    return lv;
}

// every (c, sh) this axis admits with at most max_buckets buckets, reduced to the cheapest table per level count
inline std::vector<AxisLutChoice> axis_lut_candidates(const std::vector<double>& a, int max_buckets)
{
    std::vector<AxisLutChoice> best;       // index = levels
    const int n = (int)a.size();
    // the plain bisection as a table: one bucket - the sign bit of x + c, with c lifting the axis above zero
    AxisLutChoice full;
    full.c = a.front() > 0.0 ? 0.0 : lut_hi_only(1.0 - a.front());
    full.sh = 31; full.b0 = 0; full.nbk = 1; full.win = n; full.levels = lut_levels(n);
    best.assign(full.levels + 1, full);
    for (auto& b : best) b.nbk = -1;       // not available
    best[full.levels] = full;
    if (n < 3 || n > 256) return best;
    std::vector<double> shifts;
    if (a.front() > 0.0) shifts.push_back(0.0);
    const double range = a.back() - a.front();
    if (range > 0.0 && std::isfinite(range)) {
        int e;
        (void)std::frexp(range, &e);       // range < 2^e
        for (int k = e - 12; k <= e + 1; ++k) shifts.push_back(lut_hi_only(std::ldexp(1.0, k) - a.front()));
    }
    std::vector<int> bk(n);
    for (double c : shifts) {
        if (!(a.front() + c > 0.0) || !std::isfinite(a.back() + c)) continue;
        for (int sh = 0; sh < 31; ++sh) {
            const int b0 = lut_hi32(a.front() + c) >> sh;
            const int64_t nbk = (int64_t)(lut_hi32(a.back() + c) >> sh) - b0 + 1;
            if (nbk < 1 || nbk > max_buckets) continue;
            for (int j = 0; j < n; ++j) bk[j] = lut_bucket(a[j], c, sh, b0);
            // F: nodes of one bucket that may still lie at or below x once start(b) is known (bucket 0 starts at its
            // own first node, which is a_0 <= x)
            int F = 0, run = 0;
            for (int j = 0; j < n; ++j) {
                run = (j > 0 && bk[j] == bk[j - 1]) ? run + 1 : 1;
                const int f = bk[j] == 0 ? run - 1 : run;
                F = f > F ? f : F;
            }
            AxisLutChoice ch;
            ch.c = c; ch.sh = sh; ch.b0 = b0; ch.nbk = (int)nbk; ch.win = F + 1; ch.levels = lut_levels(F + 1);
            if (ch.levels >= full.levels) continue;
            AxisLutChoice& cur = best[ch.levels];
            if (cur.nbk < 0 || ch.nbk < cur.nbk || (ch.nbk == cur.nbk && ch.win < cur.win)) cur = ch;
        }
    }
    return best;
}

// start(b) for every bucket, shifted down where the window would run past the last node (a lower start is still a
// node at or below x, and the window then ends exactly at n - 1)
inline void axis_lut_fill(const std::vector<double>& a, const AxisLutChoice& ch, uint8_t* out)
{
    const int n = (int)a.size();
    int j = 0;
    for (int b = 0; b < ch.nbk; ++b) {
        while (j < n && lut_bucket(a[j], ch.c, ch.sh, ch.b0) < b) ++j;      // j = #{ nodes with bucket < b }
        int start = j > 0 ? j - 1 : 0;
        if (start + ch.win - 1 > n - 1) start = n - ch.win;
        out[b] = (uint8_t)(start < 0 ? 0 : start);
    }
}

// One choice per axis: the fewest levels each axis can have while all tables together stay within `budget` bytes
// (start from the best of every axis, then give up one level on whichever axis frees the most bytes).
inline std::vector<AxisLutChoice> axis_lut_plan(const std::vector<const std::vector<double>*>& axes, int budget)
{
    const int na = (int)axes.size();
    std::vector<std::vector<AxisLutChoice>> cand(na);
    std::vector<int> lv(na);
    for (int a = 0; a < na; ++a) {
        cand[a] = axis_lut_candidates(*axes[a], budget);
        lv[a] = 0;
        while (cand[a][lv[a]].nbk < 0) ++lv[a];
    }
    auto total = [&]() {
        int t = 0;
        for (int a = 0; a < na; ++a) t += cand[a][lv[a]].nbk;
        return t;
    };
    while (total() > budget) {
        int pick = -1, gain = 0, next_lv = 0;
        for (int a = 0; a < na; ++a) {
            int l = lv[a] + 1;
            while (l < (int)cand[a].size() && cand[a][l].nbk < 0) ++l;
            if (l >= (int)cand[a].size()) continue;
            const int g = cand[a][lv[a]].nbk - cand[a][l].nbk;
            if (g > gain) { gain = g; pick = a; next_lv = l; }
        }
        if (pick < 0) break;               // nothing left to give up (cannot happen: the full bisection costs 1 byte)
        lv[pick] = next_lv;
    }
    std::vector<AxisLutChoice> out(na);
    for (int a = 0; a < na; ++a) out[a] = cand[a][lv[a]];
    return out;
}

}  // namespace iso
