// lpbox_capi.hip -- the C-ABI of liblpbox_hip.so (include/lpbox_hip.h): handle management, host-side index
// bookkeeping of early fixing, instance readers and result getters.  All solver arithmetic runs in the HIP kernels of
// lpbox_lp_kernels.hip; there is no CPU fallback.
//
// Reference citations: LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp,
//                      LPh   = .../cython_solver/LPboxADMMsolver.h, pxd = .../cython_solver/LPboxADMMsolver.pxd
#include "../../include/lpbox_hip.h"
#include "lpbox_lp.h"
#include "lpbox_capi_internal.h"
#include "lpbox_policy.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
thread_local int g_device = 0;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

int lpbox_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(LPBOX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct LpInstance {
    int n = 0, l = 0, nnz = 0;
    std::vector<int> colptr, rowidx;   // CSC of E, as read (LPcpp:2416-2444)
    std::vector<int> rowptr, colidx;   // CSR of the same matrix
    std::vector<int> cpos, cperm;      // storage layout: variable j sits at position cpos[j]; cperm[pos] = j
    std::vector<int> rowG;             // lanes that share the sum of row r (1,2,4,8)
    std::vector<int> col_own;          // entries of column j summed by its own lane (= its length unless the column is split)
    std::vector<int> col_help;         // [4*j + q]: entries of column j summed by lane q of its quad as a helper (0 = none)
    std::vector<int> dir_g;            // direct x-update: dense index of row r among the G rows, -1 = D row (lpbox_set_x_update)
    int nG = 0;
    struct Help { int var, first, count; };
    std::vector<Help> help_of_pos;     // storage position -> helper chunk (var < 0: none)
    std::vector<double> b, f_org;
    // early-fix bookkeeping (LPcpp:1192-1206): original index of each live variable, in compact order
    std::vector<int> left_idx;
    std::vector<int> xi_left_idx;      // live map at the time of the last l2f call (rows of x_iters)
    int xi_rows = 0;
    bool set = false;
};

template <typename Tp>
struct DevBuf {
    Tp *p = nullptr;
    size_t count = 0;
    hipError_t alloc(size_t c) {
        release();
        count = c;
        if (c == 0) return hipSuccess;
        return hipMalloc((void **)&p, c * sizeof(Tp));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; count = 0;
    }
};

}  // namespace

struct lpbox_solver {
    int flavour = LPBOX_FLAVOUR_LP, B = 0, print_info = 0, device = 0;
    SegSolver *seg = nullptr;         // flavour SEG: everything lives in the segmentation host object
    std::vector<LpInstance> inst;
    bool finalized = false, inited = false;
    int NS = 0, LS = 0, ZS = 0, T = 0, EPT = 0;
    bool colsplit = false;
    bool identity_rows = true;   // row storage index == row id (bank-aware placement off)
    bool direct = false;          // opt-in direct x-update (lpbox_set_x_update)
    int HL = 0, HLD = 0;
    size_t lds = 0, lds_direct = 0;
    bool log_on = false; int log_rows = 0; DevBuf<double> logbuf; int log_cap = 0;   // lpbox_set_log: records of the last plain call
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double kernel_ms = 0.0;
    long long launches = 0;
    DevBuf<int> rs_ptr, cs_ptr, hs_ptr, isc, ctl, left_idx, xi_rows;
    DevBuf<uint16_t> rs_col, cs_row, rid, rmeta, rgl, cmeta;
    DevBuf<int16_t> rdir;
    DevBuf<int> dng;
    DevBuf<double> x, z1, z2, b, pd, z4, f, f_org, dsc, hist, dctl, c1_init, xhist, xi_out, Hinv;
    DevBuf<uint8_t> live, newfix, live_init;
    DevBuf<unsigned long long> stamps;
    int ws_cap = 0;        // columns of the current xhist staging buffer
    int last_ws = 0;       // window length of the last l2f call
    bool xi_valid = false;
    bool record = false;   // plain loop keeps x of every iteration (lpbox_set_record; print_fix_info 2/3)
    long xi_out_stride = 0;
    int xi_out_ws = 0;
    std::vector<int> h_isc;   // host mirror, refreshed after every solver call
    std::vector<double> h_dsc;

    LpBatchDev dev() const {
        LpBatchDev d;
        d.B = B; d.NS = NS; d.LS = LS; d.ZS = ZS;
        d.rs_ptr = rs_ptr.p; d.rs_col = rs_col.p; d.cs_ptr = cs_ptr.p; d.cs_row = cs_row.p; d.hs_ptr = hs_ptr.p; d.cmeta = cmeta.p; d.rid = rid.p; d.rmeta = rmeta.p; d.rgl = rgl.p;
        d.x = x.p; d.z1 = z1.p; d.z2 = z2.p; d.b = b.p; d.pd = pd.p; d.live = live.p; d.newfix = newfix.p;
        d.z4 = z4.p; d.f = f.p; d.dsc = dsc.p; d.isc = isc.p; d.hist = hist.p;
        d.ctl = ctl.p; d.dctl = dctl.p; d.xhist = xhist.p; d.ws_cap = ws_cap; d.logbuf = nullptr; d.log_cap = 0; d.stamps = stamps.p; d.stamp_wave = getenv("LPBOX_STAMP_WAVE") ? atoi(getenv("LPBOX_STAMP_WAVE")) : 0;
        d.H = direct ? Hinv.p : nullptr; d.HL = direct ? HL : 0; d.HLD = direct ? HLD : 0; d.rdir = rdir.p; d.dng = dng.p;
        return d;
    }
};

namespace {

bool valid_handle(lpbox_t *h) { return h != nullptr && h->B > 0; }

int check_idx(lpbox_t *h, int idx) {
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (idx < 0 || idx >= h->B) return fail(LPBOX_E_BADARG, "instance index %d out of range [0,%d)", idx, h->B);
    return LPBOX_OK;
}

int use_device(lpbox_t *h) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(LPBOX_E_NODEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(h->device));
    return LPBOX_OK;
}

int refresh_scalars(lpbox_t *h) {
    h->h_isc.resize((size_t)h->B * NI_COUNT);
    h->h_dsc.resize((size_t)h->B * ND_COUNT);
    HIPCHK(hipMemcpyAsync(h->h_isc.data(), h->isc.p, h->h_isc.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_dsc.data(), h->dsc.p, h->h_dsc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LPBOX_OK;
}

// Upload the batch and choose the workgroup geometry.
int finalize(lpbox_t *h) {
    if (h->finalized) return LPBOX_OK;
    for (int i = 0; i < h->B; i++)
        if (!h->inst[i].set) return fail(LPBOX_E_STATE, "instance %d has no problem (call read_File / set_problem first)", i);
    int rc = use_device(h);
    if (rc) return rc;
    int nmax = 0, lmax = 0, zmax = 0;
    for (auto &I : h->inst) { nmax = std::max(nmax, I.n); lmax = std::max(lmax, I.l); zmax = std::max(zmax, I.nnz); }
    if (nmax > 65534 || lmax > 65534) return fail(LPBOX_E_UNSUPPORTED, "n or l exceeds the uint16 index range of the on-chip kernel");
    // workgroup geometry: 8 wavefronts (two per SIMD of the CU) with as few slots per thread as the instance allows (1, 2 or 4; the
    // 4-slot variant keeps the vectors the PCG loop never reads out of registers); beyond 2048 positions 4 wavefronts x 8 slots.
    // LPBOX_LP_THREADS overrides (tuning only).
    const int big = std::max(nmax, lmax);
    int T = 512;                          // 8 waves with short lists beat 4 waves with long ones also at n = 2000 (512 x 4, register-lean variant)
    if (const char *e = getenv("LPBOX_LP_THREADS")) { int v = atoi(e); if (v == 256 || v == 512 || v == 1024) T = v; }
    const int max_ept = T == 256 ? 8 : (T == 1024 ? 2 : 4);
    int EPT = 1;
    while (EPT < max_ept && (long)T * EPT < big) EPT *= 2;
    if ((long)T * EPT < big && T == 512) { T = 256; EPT = 1; while (EPT < 8 && (long)T * EPT < big) EPT *= 2; }
    if ((long)T * EPT < big)
        return fail(LPBOX_E_TOOLARGE, "instance with max(n,l)=%d exceeds the on-chip kernel's %d register slots", big, T * EPT);
    h->colsplit = T == 512 || T == 1024 || (T == 256 && EPT == 2);              // variants compiled with helper lists (LP_DISPATCH)
    h->T = T; h->EPT = EPT;
    h->NS = T * EPT;                       // storage positions / row-task slots per instance
    h->LS = (lmax + 31) & ~31; h->ZS = (zmax + 7) & ~7;     // LS: a whole number of 32-row bank classes
    h->lds = lp_window_lds_bytes(T, h->NS, h->LS, h->ZS);
    if (h->lds > 160 * 1024) return fail(LPBOX_E_TOOLARGE, "instance needs %zu B of LDS (> 160 KiB per CU)", h->lds);

    if (!h->stream) HIPCHK(hipStreamCreate(&h->stream));
    if (!h->ev0) { HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1)); }
    const size_t B = h->B, NS = h->NS, LS = h->LS, ZS = h->ZS;
    HIPCHK(h->rs_ptr.alloc(B * (NS + 1))); HIPCHK(h->cs_ptr.alloc(B * (NS + 1))); HIPCHK(h->hs_ptr.alloc(B * (NS + 1))); HIPCHK(h->cmeta.alloc(B * NS));
    HIPCHK(h->rs_col.alloc(B * ZS)); HIPCHK(h->cs_row.alloc(B * ZS)); HIPCHK(h->rid.alloc(B * NS)); HIPCHK(h->rmeta.alloc(B * NS)); HIPCHK(h->rgl.alloc(B * NS));
    HIPCHK(h->live_init.alloc(B * NS));
    HIPCHK(h->x.alloc(B * NS)); HIPCHK(h->z1.alloc(B * NS)); HIPCHK(h->z2.alloc(B * NS));
    HIPCHK(h->b.alloc(B * NS)); HIPCHK(h->pd.alloc(B * NS));
    HIPCHK(h->live.alloc(B * NS)); HIPCHK(h->newfix.alloc(B * NS));
    HIPCHK(h->z4.alloc(B * LS)); HIPCHK(h->f.alloc(B * LS)); HIPCHK(h->f_org.alloc(B * LS));
    HIPCHK(h->dsc.alloc(B * ND_COUNT)); HIPCHK(h->isc.alloc(B * NI_COUNT)); HIPCHK(h->hist.alloc(B * LP_HIST));
    HIPCHK(h->ctl.alloc(B * 4)); HIPCHK(h->dctl.alloc(B)); HIPCHK(h->c1_init.alloc(B));
    HIPCHK(h->left_idx.alloc(B * NS)); HIPCHK(h->xi_rows.alloc(B));
#ifdef LPBOX_STAMPS
    HIPCHK(h->stamps.alloc(B * 16)); HIPCHK(hipMemset(h->stamps.p, 0, B * 16 * sizeof(unsigned long long)));
#endif

    std::vector<int> h_rs_ptr(B * (NS + 1), 0), h_cs_ptr(B * (NS + 1), 0), h_hs_ptr(B * (NS + 1), 0), h_isc(B * NI_COUNT, 0);
    std::vector<uint16_t> h_rs_col(B * ZS, 0), h_cs_row(B * ZS, 0), h_rid(B * NS, 0xFFFF), h_rgl(B * NS, 0), h_rmeta(B * NS, 0x10), h_cmeta(B * NS, 0);
    std::vector<uint8_t> h_live(B * NS, 0);
    std::vector<double> h_b(B * NS, 0.0), h_f(B * LS, 0.0), h_c1(B, 0.0);
    const bool nosort = getenv("LPBOX_LP_NOSORT") != nullptr;
    const bool nosplit = getenv("LPBOX_LP_NOSPLIT") != nullptr;
    const int W = h->T / 64;
    // block b of 64 consecutive (sorted) items -> storage slot: slots are dealt to the waves in snake order so that every
    // wave receives a similar amount of gather work
    auto block_base = [&](int blk) {
        const int slot = blk / W, r = blk % W;
        const int wv = (slot & 1) ? (W - 1 - r) : r;
        return slot * h->T + wv * 64;
    };
    // bank-aware lane choice (round 1): O(n * 32 * row length) host work.  One slot per thread: worth < 3 % per iteration once the long
    // columns are split, and the direct x-update wants the plain row placement: off.  Four slots per thread (n > 1024): 4-5 % per
    // iteration, measured over all eight rank shards of the j=500/k=2000 stream: on.  LPBOX_LP_BANKAWARE=1 / =0 overrides either way
    // (tools/lottery.sh measures both).
    const char *ba = getenv("LPBOX_LP_BANKAWARE");
    const bool noconflict = getenv("LPBOX_LP_NOCONFLICT") != nullptr || !(ba ? atoi(ba) != 0 : h->EPT >= 4);
    for (size_t i = 0; i < B; i++) {
        LpInstance &I = h->inst[i];
        // ---- rows: G lanes share a row so that no lane walks more than ~L entries; lane g takes entries g, g+G, ... ----
        I.rowG.assign(I.l, 1);
        if (!nosplit) {
            for (int Lt = 4; Lt <= 65536; Lt++) {
                long tot = 0;
                for (int r = 0; r < I.l; r++) {
                    const int m = I.rowptr[r + 1] - I.rowptr[r];
                    int G = 1;
                    while (G < 8 && (m + G - 1) / G > Lt) G *= 2;
                    I.rowG[r] = G; tot += G;
                }
                if (tot <= (long)NS) break;
            }
        }
        std::vector<int> rorder(I.l);
        for (int r = 0; r < I.l; r++) rorder[r] = r;
        auto chain = [&](int r) { return (I.rowptr[r + 1] - I.rowptr[r] + I.rowG[r] - 1) / I.rowG[r]; };
        if (!nosort)
            std::stable_sort(rorder.begin(), rorder.end(), [&](int a, int c) {
                if (I.rowG[a] != I.rowG[c]) return I.rowG[a] > I.rowG[c];
                return chain(a) > chain(c); });
        else
            std::stable_sort(rorder.begin(), rorder.end(), [&](int a, int c) { return I.rowG[a] > I.rowG[c]; });
        struct Task { int row, g, G; };
        std::vector<Task> task_of_slot(NS, Task{-1, 0, 1});
        std::vector<int> slot_of_row(I.l, 0);          // storage slot of lane 0 of the row's task group (lanes are consecutive)
        // Blocks of 64 consecutive (sorted) tasks -> (wave, slot).  A wave walks every slot to the longest list of its 64 lanes, in chunks
        // of 4 gathers, and the phase ends when the slowest wave does; the sort is by (lanes per row, list length), so the block maxima are
        // not monotone and the snake deal left the waves of a multi-slot layout up to 35 % apart (j=500/k=2000: 24 ... 44 chunks-of-4
        // entries per wave).  Multi-slot variants therefore deal the blocks longest-first to the least loaded wave that has a free slot.
        // Where a row's task sits changes nothing in the arithmetic (a row sum is the same sum in any lane), only the time.
        std::vector<int> row_block_base;
        if (!nosort && h->EPT >= 2 && getenv("LPBOX_LP_SNAKEROWS") == nullptr) {
            long ntask = 0;
            for (int r = 0; r < I.l; r++) ntask += I.rowG[r];
            const int nb = (int)((ntask + 63) / 64);
            std::vector<int> bmax(nb, 0);
            long qq = 0;
            for (int r : rorder) { for (int g = 0; g < I.rowG[r]; g++, qq++) bmax[qq / 64] = std::max(bmax[qq / 64], chain(r)); }
            std::vector<int> border(nb);
            for (int b2 = 0; b2 < nb; b2++) border[b2] = b2;
            auto cost = [&](int b2) { return (bmax[b2] + 3) / 4 * 4; };
            std::stable_sort(border.begin(), border.end(), [&](int a, int c) { return cost(a) > cost(c); });
            std::vector<int> load(W, 0), used(W, 0);
            row_block_base.assign(nb, 0);
            for (int b2 : border) {
                int best = -1;
                for (int w = 0; w < W; w++) if (used[w] < h->EPT && (best < 0 || load[w] < load[best])) best = w;
                row_block_base[b2] = used[best] * h->T + best * 64;
                used[best]++; load[best] += cost(b2);
            }
        }
        int q = 0, max_chain = 1;
        for (int r : rorder) {
            max_chain = std::max(max_chain, chain(r));
            for (int g = 0; g < I.rowG[r]; g++, q++) {
                const int tp = nosort ? q : (row_block_base.empty() ? block_base(q / 64) : row_block_base[q / 64]) + q % 64;
                if (g == 0) slot_of_row[r] = tp;
                task_of_slot[tp] = Task{r, g, I.rowG[r]};
            }
        }
        // ---- columns: variable j -> storage position cpos[j].  Blocks of 64 by decreasing column length (stable) are dealt
        // to the waves; INSIDE a block the lane (= LDS bank class pos % 32 of the variable in the gathered vector) is chosen
        // greedily so that the 32 lanes of a half-wave gather from different banks in as many row-gather instructions as
        // possible (an instruction = the k-th list entry of the 32 row tasks of one half-wave).
        I.cperm.resize(I.n); I.cpos.resize(I.n);
        for (int j = 0; j < I.n; j++) I.cperm[j] = j;
        if (!nosort)
            std::stable_sort(I.cperm.begin(), I.cperm.end(), [&](int a, int c) {
                return I.colptr[a + 1] - I.colptr[a] > I.colptr[c + 1] - I.colptr[c]; });
        std::vector<int> var_of_pos(NS, -1);
        auto clen = [&](int j) { return j < 0 ? 0 : I.colptr[j + 1] - I.colptr[j]; };
        I.col_own.assign(I.n, 0); I.col_help.assign((size_t)4 * I.n, 0);
        for (int j = 0; j < I.n; j++) I.col_own[j] = clen(j);
        // occurrences of column j in the row-gather instructions: (half-wave group of the task slot, entry index k)
        std::vector<std::vector<std::pair<int, int>>> occ(I.n);
        const int ngrp = (int)NS / 32;
        std::vector<int> cnt((size_t)ngrp * max_chain * 32, 0);
        if (!nosort && !noconflict)
            for (int r = 0; r < I.l; r++) {
                const int G = I.rowG[r];
                for (int e = I.rowptr[r]; e < I.rowptr[r + 1]; e++) {
                    const int ee = e - I.rowptr[r];
                    occ[I.colidx[e]].push_back({(slot_of_row[r] + ee % G) / 32, ee / G});
                }
            }
        auto place_cost = [&](int j, int c) { long cost = 0; for (auto &o : occ[j]) cost += cnt[((size_t)o.first * max_chain + o.second) * 32 + c]; return cost; };
        auto place_commit = [&](int j, int p) {
            for (auto &o : occ[j]) cnt[((size_t)o.first * max_chain + o.second) * 32 + (p % 32)]++;
            I.cpos[j] = p; var_of_pos[p] = j;
        };
        const bool colsplit = h->colsplit && !nosort && getenv("LPBOX_LP_NOCOLSPLIT") == nullptr;
        if (nosort) {
            for (int qq = 0; qq < I.n; qq++) { I.cpos[I.cperm[qq]] = qq; var_of_pos[qq] = I.cperm[qq]; }
        } else if (!colsplit) {
            // Blocks of 64 by decreasing column length (stable) are dealt to the waves; INSIDE a block the lane (= LDS bank class
            // pos % 32 of the variable in the gathered vector) is chosen greedily so that the 32 lanes of a half-wave gather from
            // different banks in as many row-gather instructions as possible (an instruction = the k-th list entry of the 32 row
            // tasks of one half-wave).
            for (int blk = 0; blk * 64 < I.n; blk++) {
                bool used[64] = {false};
                const int base = block_base(blk);
                for (int qq = blk * 64; qq < std::min(I.n, blk * 64 + 64); qq++) {
                    const int j = I.cperm[qq];
                    int best = -1; long best_cost = 0;
                    for (int c = 0; c < 32; c++) {
                        if (used[c] && used[c + 32]) continue;
                        const long cost = noconflict ? 0 : place_cost(j, c);
                        if (best < 0 || cost < best_cost) { best = c; best_cost = cost; }
                    }
                    const int lane = used[best] ? best + 32 : best;
                    used[lane] = true;
                    place_commit(j, base + lane);
                }
            }
        } else {
            // One slot per thread: the per-wave issue rate of LDS gathers, not the LDS array, bounds a sparse product, so what
            // counts is the LONGEST list of a wave.  Columns are grouped in quads of adjacent lanes, one long column with three
            // short ones (long ranks ascending meet short ranks descending, so the quads of a wave look alike): the long column
            // keeps its first tau entries (tau = longest companion), the rest is dealt in consecutive chunks to the other three
            // lanes (helper lists, summed into a second accumulator and combined over the quad, lp_window_kernel cols_gather).
            const int Q = (int)NS / 4, QW = 16, CH = 8;           // quads, quads per wave, register capacity of a helper list
            const int split_bias = getenv("LPBOX_LP_SPLITBIAS") ? atoi(getenv("LPBOX_LP_SPLITBIAS")) : 2;   // cost of the quad combine, in list entries (tuning)
            auto var_of_rank = [&](int r) { return r < I.n ? I.cperm[r] : -1; };
            struct Quad { int v[4]; int tau, tail, slot; };
            std::vector<Quad> quad(Q);
            std::vector<int> quad_of_var(I.n, -1);
            auto r2 = [](int v) { return (v + 1) & ~1; };
            std::vector<int> blk_cost(Q / QW, 0);
            for (int w = 0; w < Q / QW; w++) {
                int A = 0, Bm = 0, Lm = 0;
                for (int qi = 0; qi < QW; qi++) {
                    Quad &qd = quad[w * QW + qi];
                    qd.v[0] = var_of_rank(w * QW + qi);
                    for (int t = 0; t < 3; t++) qd.v[1 + t] = var_of_rank((int)NS - 1 - (3 * (w * QW + qi) + t));
                    const int L = clen(qd.v[0]);
                    const int s1 = std::max(clen(qd.v[1]), std::max(clen(qd.v[2]), clen(qd.v[3])));
                    qd.tau = std::min(L, std::max(s1, L - 3 * CH));
                    qd.tail = L - qd.tau; qd.slot = -1;
                    A = std::max(A, std::max(qd.tau, s1)); Bm = std::max(Bm, (qd.tail + 2) / 3); Lm = std::max(Lm, std::max(L, s1));
                    for (int t = 0; t < 4; t++) if (qd.v[t] >= 0) quad_of_var[qd.v[t]] = w * QW + qi;
                }
                bool split = true;
                if (r2(A) + r2(Bm) + split_bias >= r2(Lm)) {      // splitting does not shorten this wave's longest list
                    split = false;
                    for (int qi = 0; qi < QW; qi++) { Quad &qd = quad[w * QW + qi]; qd.tau = clen(qd.v[0]); qd.tail = 0; }
                }
                blk_cost[w] = split ? (A + 3) / 4 * 4 + (Bm + 3) / 4 * 4 : (Lm + 3) / 4 * 4;
            }
            // multi-slot layouts: logical blocks of columns -> (wave, slot) longest-first to the least loaded wave, like the row tasks above
            // (own list + helper list, in chunks of 4).  Unlike the rows this moves variables to other lanes, i.e. it is part of the
            // layout the oracle mirrors through lpbox_get_layout; LPBOX_LP_SNAKECOLS=1 restores the snake deal.
            std::vector<int> col_block_base;
            if (h->EPT >= 2 && getenv("LPBOX_LP_SNAKECOLS") == nullptr) {
                const int nb = Q / QW;
                std::vector<int> border(nb);
                for (int b2 = 0; b2 < nb; b2++) border[b2] = b2;
                std::stable_sort(border.begin(), border.end(), [&](int a, int c) { return blk_cost[a] > blk_cost[c]; });
                std::vector<int> load(W, 0), usedw(W, 0);
                col_block_base.assign(nb, 0);
                for (int b2 : border) {
                    int best = -1;
                    for (int w = 0; w < W; w++) if (usedw[w] < h->EPT && (best < 0 || load[w] < load[best])) best = w;
                    col_block_base[b2] = usedw[best] * h->T + best * 64;
                    usedw[best]++; load[best] += blk_cost[b2];
                }
            }
            // lane of every column: bank-aware greedy as above, inside the wave's free quad slots / the quad's free lanes
            std::vector<char> used(NS, 0), slot_used(Q, 0);
            auto qbase = [&](int w) { return (col_block_base.empty() ? block_base(w) : col_block_base[w]) / 4; };    // first quad slot of the 64 positions that hold logical block w
            for (int qq = 0; qq < I.n; qq++) {
                const int j = I.cperm[qq];
                Quad &qd = quad[quad_of_var[j]];
                const int w = quad_of_var[j] / QW;
                int best = -1; long best_cost = 0;
                for (int t = (qd.slot >= 0 ? qd.slot : qbase(w)); t < (qd.slot >= 0 ? qd.slot + 1 : qbase(w) + QW); t++) {
                    if (qd.slot < 0 && slot_used[t]) continue;
                    for (int p = 4 * t; p < 4 * t + 4; p++) {
                        if (used[p]) continue;
                        const long cost = noconflict ? 0 : place_cost(j, p % 32);
                        if (best < 0 || cost < best_cost) { best = p; best_cost = cost; }
                    }
                }
                if (qd.slot < 0) { qd.slot = best / 4; slot_used[qd.slot] = 1; }
                used[best] = 1;
                place_commit(j, best);
            }
            for (int qd_i = 0; qd_i < Q; qd_i++) {               // quads made of holes only still need a slot (nothing is stored there)
                Quad &qd = quad[qd_i];
                if (qd.slot >= 0) continue;
                for (int t = qbase(qd_i / QW); t < qbase(qd_i / QW) + QW; t++) if (!slot_used[t]) { qd.slot = t; slot_used[t] = 1; break; }
            }
            // chunks of the tails, in lane order over the helper lanes of the quad
            I.help_of_pos.assign(NS, {-1, 0, 0});
            for (auto &qd : quad) {
                if (qd.tail <= 0 || qd.v[0] < 0) continue;
                const int jl = qd.v[0], pl = I.cpos[jl];
                I.col_own[jl] = qd.tau;
                int given = 0, hl = 0;
                for (int p = 4 * qd.slot; p < 4 * qd.slot + 4; p++) {
                    if (p == pl) continue;
                    const int c = qd.tail / 3 + (hl < qd.tail % 3 ? 1 : 0);
                    I.help_of_pos[p] = {jl, qd.tau + given, c};
                    I.col_help[(size_t)4 * jl + (p - 4 * qd.slot)] = c;
                    given += c; hl++;
                }
            }
        }
        int k = 0;
        // ---- row storage index in the gathered l-vectors (bank class rpos % 32), chosen the same way for the column gathers ----
        std::vector<int> rpos(I.l);
        for (int r = 0; r < I.l; r++) rpos[r] = r;
        if (!nosort && !noconflict) {
            h->identity_rows = false;
            int max_col = 1;
            for (int j = 0; j < I.n; j++) max_col = std::max(max_col, I.colptr[j + 1] - I.colptr[j]);
            const int ngrp = (int)NS / 32, cap = (int)LS / 32;
            std::vector<int> cnt((size_t)ngrp * 2 * max_col * 32, 0), usedc(32, 0);
            std::vector<int> order(I.l);
            for (int r = 0; r < I.l; r++) order[r] = r;
            std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return I.rowptr[a + 1] - I.rowptr[a] > I.rowptr[c + 1] - I.rowptr[c]; });
            // the column-gather instruction that reads row r for column j: (half-wave group of the reading lane, entry index in its
            // list); helper lists are separate instructions, numbered after the own lists
            auto instr_of = [&](int j, int r) {
                const int rank = (int)(std::lower_bound(I.rowidx.begin() + I.colptr[j], I.rowidx.begin() + I.colptr[j + 1], r) - (I.rowidx.begin() + I.colptr[j]));
                if (rank < I.col_own[j]) return (size_t)(I.cpos[j] / 32) * 2 * max_col + rank;
                int first = I.col_own[j];
                const int q0 = I.cpos[j] & ~3;
                for (int q = 0; q < 4; q++) {
                    const int c = I.col_help[(size_t)4 * j + q];
                    if (rank < first + c) return (size_t)((q0 + q) / 32) * 2 * max_col + max_col + (rank - first);
                    first += c;
                }
                return (size_t)0;
            };
            std::vector<size_t> ins(I.rowptr[I.l]);                 // gather instruction of every entry, row-major
            for (int r = 0; r < I.l; r++)
                for (int e = I.rowptr[r]; e < I.rowptr[r + 1]; e++) ins[e] = instr_of(I.colidx[e], r) * 32;
            // (re-choosing every row's class against all the others in further passes was measured: 77.9 / 78.1 / 78.2 us per iteration
            // of the four-slot variant with 0 / 3 / 10 passes -- nothing; one greedy pass stays)
            for (int r : order) {
                int best = -1; long best_cost = 0;
                for (int c = 0; c < 32; c++) {
                    if (usedc[c] >= cap) continue;
                    long cost = 0;
                    for (int e = I.rowptr[r]; e < I.rowptr[r + 1]; e++) cost += cnt[ins[e] + c];
                    if (best < 0 || cost < best_cost) { best = c; best_cost = cost; }
                }
                for (int e = I.rowptr[r]; e < I.rowptr[r + 1]; e++) cnt[ins[e] + best]++;
                rpos[r] = best + 32 * usedc[best]++;
            }
        }
        for (size_t p = 0; p < NS; p++) {
            h_cs_ptr[i * (NS + 1) + p] = k;
            const int j = var_of_pos[p];
            if (j < 0) continue;
            for (int e = I.colptr[j]; e < I.colptr[j] + I.col_own[j]; e++) h_cs_row[i * ZS + k++] = (uint16_t)rpos[I.rowidx[e]];
            h_b[i * NS + p] = I.b[j];
            h_live[i * NS + p] = 1;
            h_cmeta[i * NS + p] = (uint16_t)(clen(j) | (I.col_own[j] < clen(j) ? 0x8000 : 0));
        }
        h_cs_ptr[i * (NS + 1) + NS] = k;
        for (size_t p = 0; p < NS; p++) {                          // helper chunks follow the own parts in the same index pool
            h_hs_ptr[i * (NS + 1) + p] = k;
            if (I.help_of_pos.empty() || I.help_of_pos[p].var < 0) continue;
            const auto &hp = I.help_of_pos[p];
            for (int e = I.colptr[hp.var] + hp.first; e < I.colptr[hp.var] + hp.first + hp.count; e++) h_cs_row[i * ZS + k++] = (uint16_t)rpos[I.rowidx[e]];
        }
        h_hs_ptr[i * (NS + 1) + NS] = k;
        k = 0;
        for (size_t tp = 0; tp < NS; tp++) {
            h_rs_ptr[i * (NS + 1) + tp] = k;
            const Task &t = task_of_slot[tp];
            if (t.row < 0) continue;
            for (int e = I.rowptr[t.row] + t.g; e < I.rowptr[t.row + 1]; e += t.G) h_rs_col[i * ZS + k++] = (uint16_t)I.cpos[I.colidx[e]];
            h_rid[i * NS + tp] = (uint16_t)t.row;
            h_rgl[i * NS + tp] = (uint16_t)rpos[t.row];
            h_rmeta[i * NS + tp] = (uint16_t)((t.G << 4) | t.g);
        }
        h_rs_ptr[i * (NS + 1) + NS] = k;
        for (int r = 0; r < I.l; r++) h_f[i * LS + r] = I.f_org[r];
        h_isc[i * NI_COUNT + NI_N] = I.n; h_isc[i * NI_COUNT + NI_L] = I.l; h_isc[i * NI_COUNT + NI_NNZ] = I.nnz;
        h_isc[i * NI_COUNT + NI_ACTIVE] = 1;
        h_c1[i] = std::pow((double)I.n, 1.0 / 2);     // std::pow(n, 1.0/p), p = projection_lp = 2 (LPcpp:427,503)
    }
    HIPCHK(hipMemcpy(h->rmeta.p, h_rmeta.data(), h_rmeta.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->rgl.p, h_rgl.data(), h_rgl.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->live_init.p, h_live.data(), h_live.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->rs_ptr.p, h_rs_ptr.data(), h_rs_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cs_ptr.p, h_cs_ptr.data(), h_cs_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->hs_ptr.p, h_hs_ptr.data(), h_hs_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cmeta.p, h_cmeta.data(), h_cmeta.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->rs_col.p, h_rs_col.data(), h_rs_col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->cs_row.p, h_cs_row.data(), h_cs_row.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->rid.p, h_rid.data(), h_rid.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->b.p, h_b.data(), h_b.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->f_org.p, h_f.data(), h_f.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->isc.p, h_isc.data(), h_isc.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->c1_init.p, h_c1.data(), h_c1.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(h->ctl.p, 0, B * 4 * sizeof(int)));
    HIPCHK(hipMemset(h->dctl.p, 0, B * sizeof(double)));
    HIPCHK(hipMemset(h->newfix.p, 0, B * NS));
    h->finalized = true;
    return LPBOX_OK;
}

int run_window(lpbox_t *h, int iter_start, int iter_end, int l2f, bool log = false) {
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    LpBatchDev bd = h->dev();
    if (log) { bd.logbuf = h->logbuf.p; bd.log_cap = h->log_cap; }
    HIPCHK(lp_launch_window(bd, h->T, h->EPT, h->direct ? h->lds_direct : h->lds, iter_start, iter_end, l2f, h->stream, h->direct, log));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    int rc = refresh_scalars(h);     // synchronises the stream
    if (rc) return rc;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->kernel_ms += ms;
    h->launches++;
    return LPBOX_OK;
}

int set_instance(lpbox_t *h, int idx, int n, int l, int nnz, const int *colptr, const int *rowidx, const double *vals,
                 const double *b, const double *f) {
    if (h->finalized) return fail(LPBOX_E_STATE, "problem already uploaded; create a new handle to change it");
    if (n <= 0 || l <= 0 || nnz < 0 || !colptr || (!rowidx && nnz) || !b) return fail(LPBOX_E_BADARG, "bad problem arguments");
    if (n == l) return fail(LPBOX_E_UNSUPPORTED, "n == l: the reference's aliased sparse product is ill-defined here (LPcpp:103-107,150)");
    if (colptr[0] != 0 || colptr[n] != nnz) return fail(LPBOX_E_BADARG, "colptr does not span nnz");
    // validate everything BEFORE touching the instance (a rejected call leaves it as it was) and before reading rowidx / vals through colptr
    for (int j = 0; j < n; j++) {
        if (colptr[j] < 0 || colptr[j + 1] < colptr[j] || colptr[j + 1] > nnz) return fail(LPBOX_E_BADARG, "colptr not monotone inside [0, nnz]");
        for (int k = colptr[j]; k < colptr[j + 1]; k++) {
            if (rowidx[k] < 0 || rowidx[k] >= l) return fail(LPBOX_E_BADARG, "row index out of range");
            if (k > colptr[j] && rowidx[k] <= rowidx[k - 1]) return fail(LPBOX_E_BADARG, "row indices must ascend inside a column");
            if (vals && vals[k] != 1.0)
                return fail(LPBOX_E_UNSUPPORTED, "E has a stored value %g != 1; the LP kernels hold E implicitly as a 0/1 pattern", vals[k]);
        }
    }
    LpInstance &I = h->inst[idx];
    I.n = n; I.l = l; I.nnz = nnz;
    I.colptr.assign(colptr, colptr + n + 1);
    I.rowidx.assign(rowidx, rowidx + nnz);
    // CSR of E: rows in ascending column order (the order Eigen's column-major product accumulates a row in)
    I.rowptr.assign(l + 1, 0);
    for (int k = 0; k < nnz; k++) I.rowptr[rowidx[k] + 1]++;
    for (int r = 0; r < l; r++) I.rowptr[r + 1] += I.rowptr[r];
    I.colidx.assign(nnz, 0);
    std::vector<int> cur(I.rowptr.begin(), I.rowptr.end() - 1);
    for (int j = 0; j < n; j++)
        for (int k = colptr[j]; k < colptr[j + 1]; k++) I.colidx[cur[rowidx[k]]++] = j;
    I.b.assign(b, b + n);
    if (f) I.f_org.assign(f, f + l); else I.f_org.assign(l, 1.0);
    I.left_idx.resize(n);
    for (int j = 0; j < n; j++) I.left_idx[j] = j;
    I.set = true;
    return LPBOX_OK;
}

// n-vectors come back in ORIGINAL variable order (by_var = true undoes the storage permutation); l-vectors are stored by row id
int fetch_vec(lpbox_t *h, const double *pool, size_t stride, int idx, int len, std::vector<double> &out, bool by_var = true) {
    out.resize(len);
    if (len == 0) return LPBOX_OK;
    const size_t cnt = by_var ? (size_t)h->NS : (size_t)len;      // n-vectors are stored by position (NS slots incl. holes)
    std::vector<double> tmp(cnt);
    HIPCHK(hipMemcpy(tmp.data(), pool + (size_t)idx * stride, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    const LpInstance &I = h->inst[idx];
    if (by_var) for (int j = 0; j < len; j++) out[j] = tmp[I.cpos[j]];
    else out.swap(tmp);
    return LPBOX_OK;
}

int fetch_live(lpbox_t *h, int idx, std::vector<uint8_t> &out) {
    const LpInstance &I = h->inst[idx];
    std::vector<uint8_t> tmp(h->NS);
    out.resize(I.n);
    HIPCHK(hipMemcpy(tmp.data(), h->live.p + (size_t)idx * h->NS, tmp.size(), hipMemcpyDeviceToHost));
    for (int j = 0; j < I.n; j++) out[j] = tmp[I.cpos[j]];
    return LPBOX_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char *lpbox_version(void) { return "lpbox_hip 0.1 (gfx950)"; }
const char *lpbox_last_error(void) { return g_err.c_str(); }

int lpbox_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int lpbox_set_device(int device) {
    int cnt = lpbox_device_count();
    if (device < 0 || device >= cnt) return fail(LPBOX_E_NODEVICE, "device %d not available (%d visible)", device, cnt);
    g_device = device;
    return LPBOX_OK;
}

lpbox_t *lpbox_create(int flavour, int batch, int print_info) {
    if (batch <= 0 || (flavour != LPBOX_FLAVOUR_LP && flavour != LPBOX_FLAVOUR_SEG) || (flavour == LPBOX_FLAVOUR_SEG && batch != 1)) {
        fail(LPBOX_E_BADARG, "lpbox_create: unsupported flavour %d or batch %d", flavour, batch);
        return nullptr;
    }
    lpbox_t *h = new lpbox_solver();
    h->flavour = flavour; h->B = batch; h->print_info = print_info; h->device = g_device;
    if (flavour == LPBOX_FLAVOUR_SEG) h->seg = segc_create(print_info, g_device);
    else h->inst.resize(batch);
    return h;
}

void lpbox_destroy(lpbox_t *h) {
    if (!h) return;
    if (h->seg) { segc_destroy(h->seg); delete h; return; }
    if (h->finalized) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->rs_ptr.release(); h->cs_ptr.release(); h->hs_ptr.release(); h->cmeta.release(); h->isc.release(); h->ctl.release(); h->left_idx.release(); h->xi_rows.release();
    h->rs_col.release(); h->cs_row.release(); h->rid.release(); h->rmeta.release(); h->rgl.release(); h->live_init.release();
    h->x.release(); h->z1.release(); h->z2.release(); h->b.release(); h->pd.release(); h->z4.release(); h->f.release();
    h->f_org.release(); h->dsc.release(); h->hist.release(); h->dctl.release(); h->c1_init.release(); h->xhist.release();
    h->xi_out.release(); h->live.release(); h->newfix.release(); h->stamps.release(); h->logbuf.release(); h->Hinv.release(); h->rdir.release(); h->dng.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int lpbox_set_problem_lp(lpbox_t *h, int idx, int n, int l, int nnz, const int *colptr, const int *rowidx,
                         const double *vals, const double *b, const double *f) {
    if (valid_handle(h) && h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the LP flavour");
    int rc = check_idx(h, idx);
    if (rc) return rc;
    return set_instance(h, idx, n, l, nnz, colptr, rowidx, vals, b, f);
}

// readSparseMat LPcpp:2416-2444, readDenseVec :2407-2414, readFile :2446-2545
int lpbox_read_files_lp(lpbox_t *h, int idx, const char *path_C, const char *path_b, int k) {
    if (valid_handle(h) && h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the LP flavour");
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!path_C || !path_b) return fail(LPBOX_E_BADARG, "null path");
    FILE *fc = fopen(path_C, "r");
    if (!fc) return fail(LPBOX_E_IO, "cannot open %s", path_C);
    struct Trip { int r, c; double v; };
    std::vector<Trip> t;
    int row, col, max_row = 0, max_col = 0;
    double val;
    while (fscanf(fc, "%d,%d,%lf\n", &row, &col, &val) == 3) {
        if (row < 1 || col < 1) { fclose(fc); return fail(LPBOX_E_IO, "%s: indices are 1-based", path_C); }
        max_row = std::max(max_row, row); max_col = std::max(max_col, col);
        t.push_back({row - 1, col - 1, k == 2 ? -1.0 * val : val});      // :2436-2439
    }
    fclose(fc);
    if (t.empty()) return fail(LPBOX_E_IO, "%s: no 'row,col,val' triplets", path_C);
    // setFromTriplets (:2441-2443): column-major, sorted rows, duplicates summed
    std::vector<int> colptr(max_col + 1, 0);
    for (auto &e : t) colptr[e.c + 1]++;
    for (int j = 0; j < max_col; j++) colptr[j + 1] += colptr[j];
    std::vector<Trip> s(t.size());
    {
        std::vector<int> pos(colptr.begin(), colptr.end() - 1);
        for (auto &e : t) s[pos[e.c]++] = e;
    }
    std::vector<int> rowidx; std::vector<double> vals; std::vector<int> cp(max_col + 1, 0);
    for (int j = 0; j < max_col; j++) {
        std::stable_sort(s.begin() + colptr[j], s.begin() + colptr[j + 1], [](const Trip &a, const Trip &b2) { return a.r < b2.r; });
        for (int q = colptr[j]; q < colptr[j + 1]; q++) {
            if (q > colptr[j] && s[q].r == s[q - 1].r) vals.back() += s[q].v;
            else { rowidx.push_back(s[q].r); vals.push_back(s[q].v); }
        }
        cp[j + 1] = (int)rowidx.size();
    }
    FILE *fb = fopen(path_b, "r");
    if (!fb) return fail(LPBOX_E_IO, "cannot open %s", path_b);
    std::vector<double> b(max_col);
    for (int i = 0; i < max_col; i++) {
        if (fscanf(fb, "%lf\n", &b[i]) != 1) { fclose(fb); return fail(LPBOX_E_IO, "error when reading dense vector %s (entry %d)", path_b, i); }
        b[i] = -1.0 * b[i];                                                // :2520
    }
    fclose(fb);
    std::vector<double> f(max_row, 1.0);                                   // :2522
    return set_instance(h, idx, max_col, max_row, (int)rowidx.size(), cp.data(), rowidx.data(), vals.data(), b.data(), f.data());
}

int lpbox_read_file(lpbox_t *h, int idx, const char *root, int i, int k, int j) {
    std::string r = root ? root : "../cython_solver/data";                 // :2451
    char pc[1024], pb[1024];
    snprintf(pc, sizeof(pc), "%s/instance/%d_%d/instance_%d_C.txt", r.c_str(), k, j, i);   // :2492
    snprintf(pb, sizeof(pb), "%s/instance/%d_%d/instance_%d_b.txt", r.c_str(), k, j, i);   // :2494
    return lpbox_read_files_lp(h, idx, pc, pb, k);
}

int lpbox_init(lpbox_t *h) {
    if (valid_handle(h) && h->seg) return segc_init(h->seg);
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    int rc = finalize(h);
    if (rc) return rc;
    rc = use_device(h);
    if (rc) return rc;
    for (auto &I : h->inst) {
        I.left_idx.resize(I.n);
        for (int j = 0; j < I.n; j++) I.left_idx[j] = j;
        I.xi_rows = 0; I.xi_left_idx.clear();
    }
    h->xi_valid = false;
    HIPCHK(hipMemsetAsync(h->ctl.p, 0, (size_t)h->B * 4 * sizeof(int), h->stream));
    HIPCHK(lp_launch_init(h->dev(), h->T, h->EPT, h->f_org.p, h->c1_init.p, h->live_init.p, h->stream));
    rc = refresh_scalars(h);
    if (rc) return rc;
    h->inited = true;
    return 1;                                                              // LPcpp:762
}

// Upload the per-window control words and clear the x_iters staging buffer (ws columns of NS doubles per instance).
static int stage_xiters(lpbox_t *h, int ws, const std::vector<int> &h_ctl, const std::vector<double> &h_dctl,
                        const std::vector<uint8_t> *h_newfix, const std::vector<int> &h_left, const std::vector<int> &h_rows) {
    const size_t B = h->B, NS = h->NS;
    if ((double)B * ws * NS * sizeof(double) > 64e9)
        return fail(LPBOX_E_BADARG, "x history of %d iterations x %zu instances would need %.1f GB", ws, B, (double)B * ws * NS * 8 / 1e9);
    if (ws > 0 && (h->ws_cap < ws || !h->xhist.p)) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(h->xhist.alloc(B * (size_t)ws * NS));
        h->ws_cap = ws;
    }
    HIPCHK(hipMemcpyAsync(h->ctl.p, h_ctl.data(), h_ctl.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->dctl.p, h_dctl.data(), h_dctl.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h_newfix) HIPCHK(hipMemcpyAsync(h->newfix.p, h_newfix->data(), h_newfix->size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->left_idx.p, h_left.data(), h_left.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->xi_rows.p, h_rows.data(), h_rows.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (ws > 0) HIPCHK(hipMemsetAsync(h->xhist.p, 0, B * (size_t)h->ws_cap * NS * sizeof(double), h->stream));   // x_iters starts as zeros
    return LPBOX_OK;
}

int lpbox_iterate(lpbox_t *h, int iter_start, int iter_end, int *rets) {
    if (valid_handle(h) && h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the LP flavour");
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->inited) return fail(LPBOX_E_STATE, "solve_init has not been called");
    int rc = use_device(h);
    if (rc) return rc;
    const bool rec = h->record && iter_end > iter_start;
    if (rec) {                         // print_fix_info 2/3 (LPcpp:776-779,903-909): keep x of every iteration of this call
        const size_t B = h->B, NS = h->NS;
        std::vector<int> h_ctl(B * 4, 0), h_rows(B, 0), h_left(B * NS, 0);
        std::vector<double> h_dctl(B, 0.0);
        for (size_t i = 0; i < B; i++) {
            LpInstance &I = h->inst[i];
            I.xi_rows = (int)I.left_idx.size();
            I.xi_left_idx = I.left_idx;
            h_rows[i] = I.xi_rows;
            for (int q = 0; q < I.xi_rows; q++) h_left[i * NS + q] = I.cpos[I.left_idx[q]];
        }
        rc = stage_xiters(h, iter_end - iter_start, h_ctl, h_dctl, nullptr, h_left, h_rows);
        if (rc) return rc;
    } else {
        HIPCHK(hipMemsetAsync(h->ctl.p, 0, (size_t)h->B * 4 * sizeof(int), h->stream));
    }
    const bool log = h->log_on && iter_end > iter_start;
    if (log) {                         // does_log (LPcpp:1013-1067): the values of the reference's per-iteration text log, one record per iteration
        if (h->direct || !lp_log_supported(h->T, h->EPT)) return fail(LPBOX_E_UNSUPPORTED, "the iteration log is built for the default PCG kernels (512 threads per instance)");
        const size_t need = (size_t)h->B * (size_t)(iter_end - iter_start) * LP_LOG_VALS;
        if (need * sizeof(double) > (size_t)1 << 30) return fail(LPBOX_E_UNSUPPORTED, "the iteration log of this call would exceed 1 GiB");
        if (h->logbuf.count < need) HIPCHK(h->logbuf.alloc(need));
        h->log_cap = iter_end - iter_start;
        HIPCHK(hipMemsetAsync(h->logbuf.p, 0xFF, need * sizeof(double), h->stream));      // NaN = iteration not logged (it broke off, or was never reached)
    }
    rc = run_window(h, iter_start, iter_end, rec ? 2 : 0, log);
    if (rc) return rc;
    h->log_rows = log ? iter_end - iter_start : 0;
    if (rec) { h->last_ws = iter_end - iter_start; h->xi_valid = true; h->xi_out_ws = 0; }
    for (int i = 0; i < h->B; i++)
        if (rets) rets[i] = h->h_isc[(size_t)i * NI_COUNT + NI_RET];
    return h->h_isc[NI_RET];
}

int lpbox_iterate_l2f(lpbox_t *h, int iter_start, int iter_end, const double *vec, long vec_stride, const int *nums,
                      int *rets) {
    if (valid_handle(h) && h->seg) {
        int ret = 0;
        if (nums && nums[0] != 0 && vec_stride < segc_get_n(h->seg))
            return fail(LPBOX_E_BADARG, "fix vector holds %ld entries, %d live variables", vec_stride, segc_get_n(h->seg));
        int rc = segc_l2f(h->seg, iter_start, iter_end, vec, nums ? nums[0] : 0, &ret);
        if (rc < 0) return rc;
        if (rets) rets[0] = ret;
        return ret;
    }
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->inited) return fail(LPBOX_E_STATE, "solve_init has not been called");
    const int ws = iter_end - iter_start;
    if (ws > LP_XITERS_COLS) return fail(LPBOX_E_BADARG, "window of %d iterations exceeds the %d columns of x_iters (LPcpp:1113)", ws, LP_XITERS_COLS);
    int rc = use_device(h);
    if (rc) return rc;
    const size_t B = h->B, NS = h->NS;
    std::vector<int> h_ctl(B * 4, 0);
    std::vector<double> h_dctl(B, 0.0);
    std::vector<uint8_t> h_newfix;
    bool any_fix = false;
    // validate everything before touching any state
    for (size_t i = 0; i < B; i++) {
        const LpInstance &I = h->inst[i];
        const int n_live = (int)I.left_idx.size();
        const int num = nums ? nums[i] : 0;
        if (num != 0 && !h->h_isc[i * NI_COUNT + NI_ACTIVE]) return fail(LPBOX_E_BADARG, "instance %zu is parked (lpbox_set_active) but has a fix request", i);
        if (num < 0 || num > n_live) return fail(LPBOX_E_BADARG, "instance %zu: fix count %d outside [0,%d]", i, num, n_live);
        if (num != 0) {
            if (!vec) return fail(LPBOX_E_BADARG, "fix vector missing");
            if (vec_stride < n_live) return fail(LPBOX_E_BADARG, "instance %zu: fix vector holds %ld entries, %d live variables", i, vec_stride, n_live);
            const double *v = vec + (size_t)i * vec_stride;
            int cnt = 0;
            for (int q = 0; q < n_live; q++) if (v[q] == 1 || v[q] == 0) cnt++;
            if (cnt != num)                                                 // the reference would run out of bounds (LPcpp:1135-1149)
                return fail(LPBOX_E_BADARG, "instance %zu: vec fixes %d variables but num = %d", i, cnt, num);
            any_fix = true;
        }
    }
    if (any_fix) h_newfix.assign(B * NS, 0);
    std::vector<int> h_rows(B, 0);
    std::vector<int> h_left(B * NS, 0);
    for (size_t i = 0; i < B; i++) {
        LpInstance &I = h->inst[i];
        const int n_live = (int)I.left_idx.size();
        const int num = nums ? nums[i] : 0;
        if (num != 0) {                                                     // LPcpp:1124-1206 index bookkeeping
            const double *v = vec + (size_t)i * vec_stride;
            std::vector<int> keep;
            keep.reserve(n_live - num);
            for (int q = 0; q < n_live; q++) {
                const int org = I.left_idx[q];
                if (v[q] == 1) h_newfix[i * NS + I.cpos[org]] = 2;
                else if (v[q] == 0) h_newfix[i * NS + I.cpos[org]] = 1;
                else keep.push_back(org);
            }
            I.left_idx.swap(keep);
            h_ctl[i * 4 + 0] = 1; h_ctl[i * 4 + 1] = num; h_ctl[i * 4 + 2] = n_live - num;
            h_dctl[i] = std::pow((double)(n_live - num), 1.0 / 2);         // LPcpp:427 with the shrunken n
        }
        I.xi_rows = n_live - num;                                           // x_iters = Zero(n - fix_num, 500), :1113
        I.xi_left_idx = I.left_idx;
        h_rows[i] = I.xi_rows;
        for (int q = 0; q < I.xi_rows; q++) h_left[i * NS + q] = I.cpos[I.left_idx[q]];   // storage position of the q-th live variable
    }
    rc = stage_xiters(h, ws, h_ctl, h_dctl, any_fix ? &h_newfix : nullptr, h_left, h_rows);
    if (rc) return rc;
    rc = run_window(h, iter_start, iter_end, 3);    // synchronises: the host staging vectors above stay alive until here
    if (rc) return rc;
    h->last_ws = ws;
    h->xi_valid = true;
    h->xi_out_ws = 0;
    for (size_t i = 0; i < B; i++) {
        if (rets) rets[i] = h->h_isc[i * NI_COUNT + NI_RET];
    }
    return h->h_isc[NI_RET];
}

int lpbox_set_x_update(lpbox_t *h, int mode) {
    if (valid_handle(h) && h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the LP flavour");
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (mode != LPBOX_XUPDATE_PCG && mode != LPBOX_XUPDATE_DIRECT) return fail(LPBOX_E_BADARG, "x-update mode %d", mode);
    if (mode == LPBOX_XUPDATE_PCG) { h->direct = false; return LPBOX_OK; }
    int rc = finalize(h);              // the geometry decides whether the dense inverse fits
    if (rc) return rc;
    rc = use_device(h);
    if (rc) return rc;
    if (!lp_direct_supported(h->T, h->EPT))
        return fail(LPBOX_E_UNSUPPORTED, "direct x-update needs n <= 512 (this batch: %d threads x %d slots)", h->T, h->EPT);
    if (!h->identity_rows) return fail(LPBOX_E_UNSUPPORTED, "direct x-update needs the plain row placement (unset LPBOX_LP_BANKAWARE)");
    if (!h->Hinv.p) {
        // Rows with pairwise disjoint columns (D) are inverted in closed form, the rest (G) through a dense |G| x |G| inverse in LDS.
        // Greedy choice of D: rows by ascending length (the XOR "dummy item" rows of an auction are short and mutually disjoint).
        const size_t B = h->B, NS = h->NS;
        int gmax = 0;
        for (auto &I : h->inst) {
            std::vector<int> order(I.l), used(I.n, 0);
            for (int r = 0; r < I.l; r++) order[r] = r;
            std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return I.rowptr[a + 1] - I.rowptr[a] < I.rowptr[c + 1] - I.rowptr[c]; });
            std::vector<char> isD(I.l, 0);
            for (int r : order) {
                bool disjoint = true;
                for (int e = I.rowptr[r]; e < I.rowptr[r + 1] && disjoint; e++) disjoint = !used[I.colidx[e]];
                if (!disjoint) continue;
                isD[r] = 1;
                for (int e = I.rowptr[r]; e < I.rowptr[r + 1]; e++) used[I.colidx[e]] = 1;
            }
            I.dir_g.assign(I.l, -1);
            I.nG = 0;
            for (int r = 0; r < I.l; r++) if (!isD[r]) I.dir_g[r] = I.nG++;
            gmax = std::max(gmax, I.nG);
        }
        if (gmax > 128) return fail(LPBOX_E_UNSUPPORTED, "direct x-update: %d rows of E share columns (at most 128 fit the on-chip inverse)", gmax);
        // pitch = 4 (mod 32) doubles: the quads of a half-wave (8 rows x 4 consecutive columns) fall on 64 distinct LDS banks
        h->HL = std::max(gmax, 1); h->HLD = ((h->HL + 27) / 32) * 32 + 4;
        h->lds_direct = lp_window_lds_bytes(h->T, h->NS, h->LS, h->ZS, h->HL, h->HLD);
        if (h->lds_direct > 160 * 1024)
            return fail(LPBOX_E_UNSUPPORTED, "direct x-update needs %zu B of LDS (> 160 KiB per CU)", h->lds_direct);
        std::vector<uint16_t> h_rid(B * NS);
        HIPCHK(hipMemcpy(h_rid.data(), h->rid.p, h_rid.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
        std::vector<int16_t> h_rdir(B * NS, -1);
        std::vector<int> h_ng(B, 0);
        for (size_t i = 0; i < B; i++) {
            const LpInstance &I = h->inst[i];
            h_ng[i] = I.nG;
            for (size_t p = 0; p < NS; p++) { const int r = h_rid[i * NS + p]; if (r != 0xFFFF && r < I.l) h_rdir[i * NS + p] = (int16_t)I.dir_g[r]; }
        }
        HIPCHK(h->rdir.alloc(B * NS)); HIPCHK(h->dng.alloc(B));
        HIPCHK(hipMemcpy(h->rdir.p, h_rdir.data(), h_rdir.size() * sizeof(int16_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->dng.p, h_ng.data(), h_ng.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h->Hinv.alloc(B * ((size_t)h->HL * h->HLD + h->LS)));
    }
    // the saved inverses belong to whatever ran before: every instance rebuilds its own at its next x-update
    HIPCHK(hipMemset2DAsync(h->isc.p + NI_H_VALID, NI_COUNT * sizeof(int), 0, sizeof(int), (size_t)h->B, h->stream));
    h->direct = true;
    return LPBOX_OK;
}

int lpbox_set_record(lpbox_t *h, int on) {
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (h->seg) return segc_set_record(h->seg, on);
    h->record = on != 0;
    return LPBOX_OK;
}

// The values of the reference's per-iteration text log (does_log, LPh:148, LPcpp:1013-1067), opt-in: while on, every lpbox_iterate call
// leaves one record of LPBOX_LOG_VALS doubles per iteration it completed.
int lpbox_set_log(lpbox_t *h, int on) {
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (h->seg) return fail(LPBOX_E_UNSUPPORTED, "the segmentation solver of the reference writes no iteration log (SEGh:225)");
    h->log_on = on != 0;
    if (!h->log_on) h->log_rows = 0;
    return LPBOX_OK;
}

// out[r * LPBOX_LOG_VALS + k], r = 0 .. returned rows - 1: the records of instance idx from the last lpbox_iterate call, in iteration order
// (iterations that stopped the loop are not logged, as in the reference).  Value 10 is converted to seconds since the call started.
int lpbox_get_log(lpbox_t *h, int idx, double *out, int cap_rows) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the LP flavour");
    if (!out || cap_rows < 0) return fail(LPBOX_E_BADARG, "null output");
    if (h->log_rows <= 0) return 0;
    rc = use_device(h);
    if (rc) return rc;
    std::vector<double> t((size_t)h->log_rows * LP_LOG_VALS);
    HIPCHK(hipMemcpy(t.data(), h->logbuf.p + (size_t)idx * h->log_cap * LP_LOG_VALS, t.size() * sizeof(double), hipMemcpyDeviceToHost));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
    int rows = 0;
    for (int r = 0; r < h->log_rows && rows < cap_rows; r++) {
        const double *src = &t[(size_t)r * LP_LOG_VALS];
        if (src[11] != src[11]) continue;                            // NaN: not logged
        double *dst = out + (size_t)rows * LP_LOG_VALS;
        for (int k = 0; k < LP_LOG_VALS; k++) dst[k] = src[k];
        dst[10] = src[10] / (1e3 * khz);
        rows++;
    }
    return rows;
}

int lpbox_seg_get_x_history(lpbox_t *h, int first, int count, double *out) {
    if (!valid_handle(h) || !h->seg) return fail(LPBOX_E_BADHANDLE, "bad handle (segmentation flavour only)");
    return segc_get_x_history(h->seg, first, count, out);
}

int lpbox_policy_layout(int tokens, long *weight_halves, long *const_floats) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (weight_halves) *weight_halves = 2L * POLICY_FRAGS_PER_LAYER * 512;
    if (const_floats) *const_floats = POLICY_OFF_LAYER(tokens) + 2L * POLICY_LAYER_CONSTS;
    return LPBOX_OK;
}

int lpbox_policy_encode_f16(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                            const void *weights_dev, const float *consts_dev, void *out_dev, void *hip_stream) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (rows < 0 || tok_stride < 1) return fail(LPBOX_E_BADARG, "bad rows / token stride");
    if (rows == 0) return LPBOX_OK;
    if (!x_dev || !row_off_dev || !weights_dev || !consts_dev || !out_dev) return fail(LPBOX_E_BADARG, "null device pointer");
    PolicyArgs pa;
    pa.x = x_dev; pa.row_off = row_off_dev; pa.rows = rows; pa.tok_stride = tok_stride;
    pa.weights = weights_dev; pa.consts = consts_dev; pa.out = out_dev;
    HIPCHK(policy_launch_body(pa, tokens, (hipStream_t)hip_stream));
    return LPBOX_OK;
}

int lpbox_policy_f32frag_layout(int tokens, long *weight_floats, long *const_floats) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (weight_floats) *weight_floats = policy_f32frag_floats();
    if (const_floats) *const_floats = POLICY_OFF_LAYER(tokens) + 2L * POLICY_LAYER_CONSTS;
    return LPBOX_OK;
}

int lpbox_policy_encode_f32(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                            const float *weights_dev, const float *consts_dev, float *out_dev, void *hip_stream) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (rows < 0 || tok_stride < 1) return fail(LPBOX_E_BADARG, "bad rows / token stride");
    if (rows == 0) return LPBOX_OK;
    if (!x_dev || !row_off_dev || !weights_dev || !consts_dev || !out_dev) return fail(LPBOX_E_BADARG, "null device pointer");
    PolicyArgs pa;
    pa.x = x_dev; pa.row_off = row_off_dev; pa.rows = rows; pa.tok_stride = tok_stride;
    pa.weights = weights_dev; pa.consts = consts_dev; pa.out = out_dev;
    HIPCHK(policy_launch_body_f32(pa, tokens, (hipStream_t)hip_stream));
    return LPBOX_OK;
}

int lpbox_policy_f32_layout(int tokens, long *weight_floats) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (weight_floats) *weight_floats = policy_f32_weight_floats(tokens);
    return LPBOX_OK;
}

int lpbox_policy_score_f32(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                           const float *weights_dev, float *sigmoid_dev, float *logit_dev, void *hip_stream) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (rows < 0 || tok_stride < 1) return fail(LPBOX_E_BADARG, "bad rows / token stride");
    if (rows == 0) return LPBOX_OK;
    if (!x_dev || !row_off_dev || !weights_dev || !sigmoid_dev) return fail(LPBOX_E_BADARG, "null device pointer");
    HIPCHK(policy_launch_f32(x_dev, row_off_dev, rows, tokens, tok_stride, weights_dev, sigmoid_dev, logit_dev, 0.f, 0.f, 0.f, nullptr, (hipStream_t)hip_stream));
    return LPBOX_OK;
}

int lpbox_policy_rescore_f32(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                             const float *weights_dev, float *sigmoid_dev, float band, float thr_hi, float thr_lo,
                             unsigned long long *count_dev, void *hip_stream) {
    if (tokens != 20 && tokens != 5) return fail(LPBOX_E_BADARG, "tokens must be 20 (LP) or 5 (segmentation)");
    if (rows < 0 || tok_stride < 1 || !(band > 0.f)) return fail(LPBOX_E_BADARG, "bad rows / token stride / band");
    if (rows == 0) return LPBOX_OK;
    if (!x_dev || !row_off_dev || !weights_dev || !sigmoid_dev) return fail(LPBOX_E_BADARG, "null device pointer");
    HIPCHK(policy_launch_f32(x_dev, row_off_dev, rows, tokens, tok_stride, weights_dev, sigmoid_dev, nullptr, band, thr_hi, thr_lo, count_dev, (hipStream_t)hip_stream));
    return LPBOX_OK;
}

int lpbox_set_active(lpbox_t *h, const int *active) {
    if (!valid_handle(h) || h->seg) return fail(LPBOX_E_BADHANDLE, "bad handle (LP flavour only)");
    if (!h->inited) return fail(LPBOX_E_STATE, "solve_init has not been called");
    int rc = use_device(h);
    if (rc) return rc;
    std::vector<int> a(h->B, 1);
    for (int i = 0; i < h->B && active; i++) a[i] = active[i] != 0;
    HIPCHK(hipMemcpy2D(h->isc.p + NI_ACTIVE, NI_COUNT * sizeof(int), a.data(), sizeof(int), sizeof(int), h->B, hipMemcpyHostToDevice));
    for (int i = 0; i < h->B; i++) h->h_isc[(size_t)i * NI_COUNT + NI_ACTIVE] = a[i];
    return LPBOX_OK;
}

int lpbox_get_n(lpbox_t *h, int idx) {
    if (valid_handle(h) && h->seg) return segc_get_n(h->seg);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return h->inst[idx].n;
    return h->h_isc[(size_t)idx * NI_COUNT + NI_NLIVE];
}

int lpbox_get_org_n(lpbox_t *h, int idx) {
    if (valid_handle(h) && h->seg) return segc_get_org_n(h->seg);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    return h->inst[idx].n;
}

int lpbox_get_l(lpbox_t *h, int idx) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    return h->inst[idx].l;
}

int lpbox_get_iter(lpbox_t *h, int idx) {
    if (valid_handle(h) && h->seg) return segc_get_iter(h->seg);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return 0;
    return h->h_isc[(size_t)idx * NI_COUNT + NI_ITER];
}

int lpbox_get_x_iters(lpbox_t *h, int idx, int ws, double *out) {
    if (valid_handle(h) && h->seg) return segc_get_x_iters(h->seg, ws, out);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->xi_valid) return fail(LPBOX_E_STATE, "solve_iter_l2f has not been called");
    const int ws_max = std::max(LP_XITERS_COLS, h->ws_cap);      // x_iters has 500 columns (LPcpp:1113); a recorded plain window may be longer
    if (ws < 0 || ws > ws_max) return fail(LPBOX_E_BADARG, "ws = %d outside [0,%d]", ws, ws_max);
    const int rows = h->inst[idx].xi_rows;
    if (!out || rows == 0 || ws == 0) return rows;
    rc = use_device(h);
    if (rc) return rc;
    const int wsd = std::min(ws, h->ws_cap);          // columns beyond the staged window stay zero, like the reference's matrix
    if (h->xi_out_ws != wsd) {                         // pack the whole batch once per (call, ws)
        const long stride = (long)h->NS * wsd;
        if (h->xi_out.count < (size_t)h->B * stride) HIPCHK(h->xi_out.alloc((size_t)h->B * stride));
        HIPCHK(lp_launch_pack_xiters(h->dev(), h->left_idx.p, h->xi_rows.p, wsd, h->xi_out.p, stride, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->xi_out_ws = wsd; h->xi_out_stride = stride;
    }
    if (wsd == ws) {
        HIPCHK(hipMemcpy(out, h->xi_out.p + (size_t)idx * h->xi_out_stride, sizeof(double) * (size_t)rows * ws, hipMemcpyDeviceToHost));
    } else {
        std::vector<double> tmp((size_t)rows * wsd);
        HIPCHK(hipMemcpy(tmp.data(), h->xi_out.p + (size_t)idx * h->xi_out_stride, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < ws; c++) out[(size_t)r * ws + c] = c < wsd ? tmp[(size_t)r * wsd + c] : 0.0;
    }
    return rows;
}

int lpbox_get_x_iters_device(lpbox_t *h, int ws, void **dev_ptr, long *stride_doubles) {
    if (valid_handle(h) && h->seg) return segc_get_x_iters_device(h->seg, ws, dev_ptr, stride_doubles);
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->xi_valid) return fail(LPBOX_E_STATE, "solve_iter_l2f has not been called");
    if (ws <= 0 || ws > h->ws_cap) return fail(LPBOX_E_BADARG, "ws = %d outside (0,%d] (the last window)", ws, h->ws_cap);
    int rc = use_device(h);
    if (rc) return rc;
    if (h->xi_out_ws != ws) {
        const long stride = (long)h->NS * ws;
        if (h->xi_out.count < (size_t)h->B * stride) HIPCHK(h->xi_out.alloc((size_t)h->B * stride));
        HIPCHK(lp_launch_pack_xiters(h->dev(), h->left_idx.p, h->xi_rows.p, ws, h->xi_out.p, stride, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->xi_out_ws = ws; h->xi_out_stride = stride;
    }
    if (dev_ptr) *dev_ptr = h->xi_out.p;
    if (stride_doubles) *stride_doubles = h->xi_out_stride;
    return LPBOX_OK;
}

int lpbox_get_x_sol(lpbox_t *h, int idx, double *out) {
    if (valid_handle(h) && h->seg) return segc_get_x_sol(h->seg, out);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !out) return fail(LPBOX_E_STATE, "not initialised");
    rc = use_device(h);
    if (rc) return rc;
    std::vector<double> x; std::vector<uint8_t> live;
    if ((rc = fetch_vec(h, h->x.p, h->NS, idx, h->inst[idx].n, x))) return rc;
    if ((rc = fetch_live(h, idx, live))) return rc;
    for (int j = 0; j < h->inst[idx].n; j++) out[j] = live[j] ? (x[j] >= 0.5 ? 1.0 : 0.0) : x[j];   // LPcpp:1648-1665
    return h->inst[idx].n;
}

int lpbox_get_final_x_sol(lpbox_t *h, int idx, double *out) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !out) return fail(LPBOX_E_STATE, "not initialised");
    rc = use_device(h);
    if (rc) return rc;
    std::vector<double> x;
    if ((rc = fetch_vec(h, h->x.p, h->NS, idx, h->inst[idx].n, x))) return rc;
    const auto &li = h->inst[idx].left_idx;
    for (size_t q = 0; q < li.size(); q++) out[q] = x[li[q]];              // LPcpp:1668-1685: the live (compacted) x_sol
    return (int)li.size();
}

int lpbox_cal_obj(lpbox_t *h, int idx, double *out) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !out) return fail(LPBOX_E_STATE, "not initialised");
    const double *d = &h->h_dsc[(size_t)idx * ND_COUNT];
    const int n_live = h->h_isc[(size_t)idx * NI_COUNT + NI_NLIVE];
    *out = n_live != 0 ? d[ND_SUM_FIX_OBJ] + d[ND_CUR_OBJ] : d[ND_SUM_FIX_OBJ];   // LPcpp:1630-1642
    return LPBOX_OK;
}

int lpbox_cur_bin_obj(lpbox_t *h, int idx, double *out) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !out) return fail(LPBOX_E_STATE, "not initialised");
    *out = h->h_dsc[(size_t)idx * ND_COUNT + ND_CUR_OBJ];
    return LPBOX_OK;
}

int lpbox_get_problem_lp(lpbox_t *h, int idx, int *n, int *l, int *nnz, int *colptr, int *rowidx, double *b, double *f) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    const LpInstance &I = h->inst[idx];
    if (!I.set) return fail(LPBOX_E_STATE, "instance %d has no problem (call read_File / set_problem first)", idx);
    if (n) *n = I.n;
    if (l) *l = I.l;
    if (nnz) *nnz = I.nnz;
    if (colptr) std::copy(I.colptr.begin(), I.colptr.end(), colptr);
    if (rowidx) std::copy(I.rowidx.begin(), I.rowidx.end(), rowidx);
    if (b) std::copy(I.b.begin(), I.b.end(), b);
    if (f) std::copy(I.f_org.begin(), I.f_org.end(), f);
    return LPBOX_OK;
}

int lpbox_check_infeasible_lpbox(lpbox_t *h, int idx) {                     // LPcpp:1577-1591: rows of the CURRENT E with (E x)_i > 1
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return fail(LPBOX_E_STATE, "not initialised");
    rc = use_device(h);
    if (rc) return rc;
    const LpInstance &I = h->inst[idx];
    if (I.left_idx.empty()) return 0;
    std::vector<double> x; std::vector<uint8_t> live;
    if ((rc = fetch_vec(h, h->x.p, h->NS, idx, I.n, x))) return rc;
    if ((rc = fetch_live(h, idx, live))) return rc;
    int inf = 0;
    for (int r = 0; r < I.l; r++) {
        double s = 0.0;
        for (int k = I.rowptr[r]; k < I.rowptr[r + 1]; k++) if (live[I.colidx[k]]) s += 1.0 * x[I.colidx[k]];
        if (!(s <= 1.0)) inf++;
    }
    return inf;
}

int lpbox_check_infeasible_l2f(lpbox_t *h, int idx) {                       // LPcpp:1593-1612: ORIGINAL E times the rounded full-length x
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return fail(LPBOX_E_STATE, "not initialised");
    const LpInstance &I = h->inst[idx];
    std::vector<double> sol(I.n);
    rc = lpbox_get_x_sol(h, idx, sol.data());
    if (rc < 0) return rc;
    int inf = 0;
    for (int r = 0; r < I.l; r++) {
        double s = 0.0;
        for (int k = I.rowptr[r]; k < I.rowptr[r + 1]; k++) s += 1.0 * sol[I.colidx[k]];
        if (!(s <= 1.0)) inf++;
    }
    return inf;
}

int lpbox_get_config(lpbox_t *h, int *threads, int *elems_per_thread, int *lds_bytes) {
    if (valid_handle(h) && h->seg) return segc_get_config(h->seg, threads, elems_per_thread, lds_bytes);
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    int rc = finalize(h);
    if (rc) return rc;
    if (threads) *threads = h->T;
    if (elems_per_thread) *elems_per_thread = h->EPT;
    if (lds_bytes) *lds_bytes = (int)h->lds;
    return LPBOX_OK;
}

int lpbox_get_layout(lpbox_t *h, int idx, int *pos_of_var) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    rc = finalize(h);
    if (rc) return rc;
    if (!pos_of_var) return fail(LPBOX_E_BADARG, "null output");
    const LpInstance &I = h->inst[idx];
    for (int j = 0; j < I.n; j++) pos_of_var[j] = I.cpos[j];
    return I.n;
}

int lpbox_get_row_split(lpbox_t *h, int idx, int *lanes_of_row) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    rc = finalize(h);
    if (rc) return rc;
    if (!lanes_of_row) return fail(LPBOX_E_BADARG, "null output");
    const LpInstance &I = h->inst[idx];
    for (int r = 0; r < I.l; r++) lanes_of_row[r] = I.rowG[r];
    return I.l;
}

int lpbox_get_col_split(lpbox_t *h, int idx, int *own, int *help4) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    rc = finalize(h);
    if (rc) return rc;
    if (!own || !help4) return fail(LPBOX_E_BADARG, "null output");
    const LpInstance &I = h->inst[idx];
    for (int j = 0; j < I.n; j++) own[j] = I.col_own[j];
    for (int k = 0; k < 4 * I.n; k++) help4[k] = I.col_help[k];
    return I.n;
}

int lpbox_get_direct_rows(lpbox_t *h, int idx, int *gidx_of_row) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the LP flavour");
    const LpInstance &I = h->inst[idx];
    if ((int)I.dir_g.size() != I.l) return fail(LPBOX_E_STATE, "lpbox_set_x_update(LPBOX_XUPDATE_DIRECT) has not been called");
    if (!gidx_of_row) return fail(LPBOX_E_BADARG, "null output");
    for (int r = 0; r < I.l; r++) gidx_of_row[r] = I.dir_g[r];
    return I.nG;
}

int lpbox_get_counters(lpbox_t *h, int idx, long long *outer_iters, long long *pcg_iters) {
    if (valid_handle(h) && h->seg) return segc_get_counters(h->seg, outer_iters, pcg_iters);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return fail(LPBOX_E_STATE, "not initialised");
    if (outer_iters) *outer_iters = h->h_isc[(size_t)idx * NI_COUNT + NI_OUTER_TOTAL];
    if (pcg_iters) *pcg_iters = h->h_isc[(size_t)idx * NI_COUNT + NI_PCG_TOTAL];
    return LPBOX_OK;
}

int lpbox_get_stop(lpbox_t *h, int idx, int *reason, int *plain_iter_plus1) {
    if (valid_handle(h) && h->seg) return segc_get_stop(h->seg, reason, plain_iter_plus1);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return fail(LPBOX_E_STATE, "not initialised");
    if (reason) *reason = h->h_isc[(size_t)idx * NI_COUNT + NI_STOP];
    if (plain_iter_plus1) *plain_iter_plus1 = h->h_isc[(size_t)idx * NI_COUNT + NI_PLAIN_ITER_P1];
    return LPBOX_OK;
}

int lpbox_kernel_time(lpbox_t *h, double *ms_total, long long *launches, int reset) {
    if (valid_handle(h) && h->seg) return segc_kernel_time(h->seg, ms_total, launches, reset);
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (ms_total) *ms_total = h->kernel_ms;
    if (launches) *launches = h->launches;
    if (reset) { h->kernel_ms = 0.0; h->launches = 0; }
    return LPBOX_OK;
}

int lpbox_debug_get_vec(lpbox_t *h, int idx, const char *name, double *out, int cap) {
    if (valid_handle(h) && h->seg) return segc_debug_vec(h->seg, name, out, cap);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !name || !out) return fail(LPBOX_E_STATE, "not initialised");
    rc = use_device(h);
    if (rc) return rc;
    const LpInstance &I = h->inst[idx];
    const double *pool = nullptr; size_t stride = h->NS; int len = I.n; bool by_var = true;
    if (!strcmp(name, "x")) pool = h->x.p;
    else if (!strcmp(name, "z1")) pool = h->z1.p;
    else if (!strcmp(name, "z2")) pool = h->z2.p;
    else if (!strcmp(name, "b")) pool = h->b.p;
    else if (!strcmp(name, "pd")) pool = h->pd.p;
    else if (!strcmp(name, "z4")) { pool = h->z4.p; stride = h->LS; len = I.l; by_var = false; }
    else if (!strcmp(name, "f")) { pool = h->f.p; stride = h->LS; len = I.l; by_var = false; }
    else if (!strcmp(name, "live")) {
        std::vector<uint8_t> live;
        if ((rc = fetch_live(h, idx, live))) return rc;
        if (I.n > cap) return fail(LPBOX_E_BADARG, "buffer too small");
        for (int j = 0; j < I.n; j++) out[j] = live[j];
        return I.n;
    } else return fail(LPBOX_E_BADARG, "unknown vector '%s'", name);
    if (len > cap) return fail(LPBOX_E_BADARG, "buffer too small");
    std::vector<double> v;
    if ((rc = fetch_vec(h, pool, stride, idx, len, v, by_var))) return rc;
    memcpy(out, v.data(), sizeof(double) * (size_t)len);
    return len;
}

// diagnostic build only: the 16 phase counters (shader cycles of wave 0) of instance idx from the last launch
int lpbox_debug_get_stamps(lpbox_t *h, int idx, unsigned long long *out16) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->stamps.p) return fail(LPBOX_E_UNSUPPORTED, "library built without LPBOX_STAMPS");
    HIPCHK(hipMemcpy(out16, h->stamps.p + (size_t)idx * 16, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return LPBOX_OK;
}

int lpbox_debug_get_scalar(lpbox_t *h, int idx, const char *name, double *out) {
    if (valid_handle(h) && h->seg) return segc_debug_scalar(h->seg, name, out);
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !name || !out) return fail(LPBOX_E_STATE, "not initialised");
    const double *d = &h->h_dsc[(size_t)idx * ND_COUNT];
    const int *q = &h->h_isc[(size_t)idx * NI_COUNT];
    struct { const char *n; double v; } tab[] = {
        {"rho1", d[ND_RHO1]}, {"rho2", d[ND_RHO2]}, {"rho4", d[ND_RHO4]}, {"prev_rho1", d[ND_PREV_RHO1]},
        {"prev_rho4", d[ND_PREV_RHO4]}, {"gamma", d[ND_GAMMA]}, {"dI", d[ND_DI]}, {"rho4Et", d[ND_R4ET]},
        {"std_obj", d[ND_STD_OBJ]}, {"cur_obj", d[ND_CUR_OBJ]}, {"sum_fix_obj", d[ND_SUM_FIX_OBJ]},
        {"best_bin_obj", d[ND_BEST_BIN_OBJ]}, {"cvg1", d[ND_CVG1]}, {"cvg2", d[ND_CVG2]}, {"obj_val", d[ND_OBJ_VAL]},
        {"rhoUpdated", (double)q[NI_RHO_UPDATED]}, {"last_pcg", (double)q[NI_LAST_PCG]},
    };
    for (auto &e : tab) if (!strcmp(e.n, name)) { *out = e.v; return LPBOX_OK; }
    return fail(LPBOX_E_BADARG, "unknown scalar '%s'", name);
}

// ---- segmentation flavour (SEG pxd = Segmentation/Segmentation/cython/src/LPboxADMMsolver.pxd) ----
static int seg_handle(lpbox_t *h) {
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->seg) return fail(LPBOX_E_STATE, "this entry point belongs to the segmentation flavour");
    return LPBOX_OK;
}

int lpbox_set_problem_bqp(lpbox_t *h, int n, int nnz, const int *rowptr, const int *colidx, const double *vals,
                          const double *b, double c, int rows, int cols) {
    int rc = seg_handle(h);
    if (rc) return rc;
    return segc_set_problem(h->seg, n, nnz, rowptr, colidx, vals, b, c, rows, cols);
}

int lpbox_seg_set_image(lpbox_t *h, const unsigned char *gray, int rows, int cols, int num_nodes) {
    int rc = seg_handle(h);
    if (rc) return rc;
    return segc_set_image(h->seg, gray, rows, cols, num_nodes);
}

int lpbox_seg_legacy(lpbox_t *h, int *energy) {
    int rc = seg_handle(h);
    if (rc) return rc;
    return segc_legacy(h->seg, energy);
}

int lpbox_seg_legacy_batch(lpbox_t **hs, int count, int *energies) {
    if (!hs || count <= 0) return fail(LPBOX_E_BADARG, "empty batch");
    std::vector<SegSolver *> ss(count);
    for (int i = 0; i < count; i++) {
        int rc = seg_handle(hs[i]);
        if (rc) return rc;
        ss[i] = hs[i]->seg;
    }
    return segc_legacy_batch(ss.data(), count, energies);
}

int lpbox_seg_get_obj(lpbox_t *h, double *out) {
    int rc = seg_handle(h);
    if (rc) return rc;
    return segc_get_obj(h->seg, out);
}

int lpbox_seg_get_shape(lpbox_t *h, int *rows, int *cols) {
    int rc = seg_handle(h);
    if (rc) return rc;
    return segc_get_shape(h->seg, rows, cols);
}

int lpbox_seg_get_problem(lpbox_t *h, int *n, int *nnz, int *rowptr, int *colidx, double *vals, double *b, double *c) {
    int rc = seg_handle(h);
    if (rc) return rc;
    return segc_get_problem(h->seg, n, nnz, rowptr, colidx, vals, b, c);
}

}  // extern "C"
