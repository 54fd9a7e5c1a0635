// ceg_rows.h -- the row-wise walk over the k-vectors of the Ewald sum, shared by ceg_recip.hip (row f2) and ceg_mc.hip (config 5).
//
// The reference stores its k-vectors as ROWS (kspace.kindices, src/ewald.jl:213-236): fixed (j, k), i = i0 .. i1, and its loops
// (ewald_main_loop! :158-176, update_sums! :660-684) form Eiky[j] Eikz[k] q once per row and atom and step along i.  Here the
// flat k-vector list is regrouped into such rows, the rows are cut into segments of <= SEG k-vectors and the segments are dealt,
// longest first, to the 64 lanes of a wave in rounds; a lane forms Ey[j] Ez[k] q ONCE per segment and atom (the charge rides on
// the z table) and steps along i with Ex[i+1] = Ex[i] Ex[1] -- per k-vector and atom 8 FMAs and no LDS access.
// Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

namespace ceg_rows {

constexpr int SEG = 10;             // k-vectors per segment (structure-factor accumulators a lane holds); the switches list 1..SEG

// One round for segments of LEN k-vectors: the structure factor of the `natoms` atoms of the tables at (i0 + s, j, k), s < LEN,
// handed to sink(s, re, im).  tab: [natoms][tab_stride] double2, entry t in [0, nxp) the x table (m = 0..kx), then y (m = -ky..ky,
// jj = j + ky), then z (kk = k + kz) WITH THE CHARGE of the atom as a factor.
template <int LEN, class Sink>
__device__ __forceinline__ void round_sf(int natoms, const double2* tab, int tab_stride, int nxp, int nyp, int i0, int jj, int kk, Sink&& sink)
{
    double sr[LEN], si[LEN];
#pragma unroll
    for (int s = 0; s < LEN; ++s) sr[s] = si[s] = 0.0;
    for (int a = 0; a < natoms; ++a) {
        const double2* ta = tab + a * tab_stride;
        const double2 ey = ta[nxp + jj], ez = ta[nxp + nyp + kk], e1 = ta[nxp > 1 ? 1 : 0];
        double2 ex = ta[i0];
        const double cr = ey.x * ez.x - ey.y * ez.y, ci = ey.x * ez.y + ey.y * ez.x;      // c*Eiky*Eikz: the z table carries the charge
#pragma unroll
        for (int s = 0; s < LEN; ++s) {
            sr[s] += ex.x * cr - ex.y * ci;
            si[s] += ex.x * ci + ex.y * cr;
            if (s + 1 < LEN) {
                const double nx = ex.x * e1.x - ex.y * e1.y;
                ex.y = ex.x * e1.y + ex.y * e1.x;
                ex.x = nx;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < LEN; ++s) sink(s, sr[s], si[s]);
}

// dispatch on the round length L (wave-uniform): one branch-free body per length, so that the loads of a round are scheduled
// ahead of its arithmetic
template <class Sink>
__device__ __forceinline__ void round_dispatch(int L, int natoms, const double2* tab, int tab_stride, int nxp, int nyp, int i0, int jj, int kk, Sink&& sink)
{
    switch (L) {
    case 1: round_sf<1>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 2: round_sf<2>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 3: round_sf<3>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 4: round_sf<4>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 5: round_sf<5>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 6: round_sf<6>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 7: round_sf<7>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 8: round_sf<8>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    case 9: round_sf<9>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    default: round_sf<SEG>(natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, sink); break;
    }
}

// Regroup the flat k-vector list: rows of consecutive i at fixed (j, k), cut into segments of <= seg k-vectors of nearly equal length,
// sorted by length (longest first) and dealt to the lanes in rounds of 64 -- the segments of one round have nearly the same length,
// so a round that runs to its longest segment wastes little.
//  desc[r * 64 + lane]: segment of `lane` in round r: i0 | (j + ky) << 9 | (k + kz) << 18 | L_r << 27 (nine bits each; L_r = longest
//                       segment of the round, the same in every lane)
//  slot_of[q]:          plane index (slot * 64 + lane) of k-vector q; the slots of round r are the L_r following those of round r - 1
//  ns:                  the padded slot count
struct Layout {
    int nrounds = 0, ns = 0;
    std::vector<int32_t> desc;
    std::vector<int64_t> slot_of;
};

inline Layout build_layout(const int32_t* ijk, int64_t nk, const int32_t ks[3], int seg)
{
    std::map<std::pair<int, int>, std::vector<std::pair<int, int64_t>>> rows;        // (j, k) -> (i, q)
    for (int64_t q = 0; q < nk; ++q) rows[{ijk[3 * q + 1], ijk[3 * q + 2]}].push_back({ijk[3 * q], q});
    struct Seg { int j, k, i0, len; std::vector<int64_t> q; };
    std::vector<Seg> segs;
    for (auto& kv : rows) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end());
        size_t b = 0;
        while (b < v.size()) {
            size_t e = b + 1;
            while (e < v.size() && v[e].first == v[e - 1].first + 1) ++e;          // run of consecutive i (a repeated i starts a new run)
            const int len = (int)(e - b), parts = (len + seg - 1) / seg;
            size_t at = b;
            for (int part = 0; part < parts; ++part) {
                const int l = len / parts + (part < len % parts ? 1 : 0);
                Seg s{kv.first.first, kv.first.second, v[at].first, l, {}};
                for (int t = 0; t < l; ++t) s.q.push_back(v[at + t].second);
                segs.push_back(std::move(s));
                at += l;
            }
            b = e;
        }
    }
    std::stable_sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.len > b.len; });
    Layout out;
    out.nrounds = (int)((segs.size() + 63) / 64);
    out.desc.assign((size_t)out.nrounds * 64, 0);
    out.slot_of.assign((size_t)nk, 0);
    int slot = 0;
    for (int r = 0; r < out.nrounds; ++r) {
        const int L = segs[(size_t)r * 64].len;
        for (int l = 0; l < 64; ++l) {
            const size_t si = (size_t)r * 64 + l;
            const int lane = (r & 1) ? 63 - l : l;
            int32_t d = (ks[1] << 9) | (ks[2] << 18);                              // padding: j = k = i0 = 0, all constants zero
            if (si < segs.size()) {
                const Seg& s = segs[si];
                d = s.i0 | ((s.j + ks[1]) << 9) | ((s.k + ks[2]) << 18);
                for (int t = 0; t < s.len; ++t) out.slot_of[(size_t)s.q[t]] = (int64_t)(slot + t) * 64 + lane;
            }
            out.desc[(size_t)r * 64 + lane] = d | (L << 27);
        }
        slot += L;
    }
    out.ns = slot;
    return out;
}

// cost model of a layout for the choice of the segment length: per round and atom one Ey Ez q product + table reads (~14 FP64
// instructions' worth), per slot 8 FMAs per atom + 5 for the energy; two atoms assumed
inline double layout_cost(const Layout& l) { return 2.0 * 14.0 * l.nrounds + (2.0 * 8.0 + 5.0) * l.ns; }

inline Layout choose_layout(const int32_t* kvec_ijk, int64_t nk, const int32_t ks[3])
{
    Layout best;
    if (nk > 0) {
        best = build_layout(kvec_ijk, nk, ks, SEG);
        for (int seg = SEG - 1; seg >= 3; --seg) {                 // shorter segments can fill the last round better
            Layout l = build_layout(kvec_ijk, nk, ks, seg);
            if (layout_cost(l) < layout_cost(best)) best = std::move(l);
        }
    }
    return best;
}

}  // namespace ceg_rows
