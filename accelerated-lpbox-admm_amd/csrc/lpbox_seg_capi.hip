// lpbox_seg_capi.hip -- host side of the SEGMENTATION flavour behind the C-ABI: image -> (A, b, c) cost construction
// (SEGcpp:46-248, :705-756), device buffers, the graph-replayed iteration driver, early-fix index bookkeeping and getters.
// All solver arithmetic runs in lpbox_seg_kernels.hip; there is no CPU fallback.
#include "../../include/lpbox_hip.h"
#include "lpbox_capi_internal.h"
#include "lpbox_seg.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return lpbox_fail(LPBOX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {
template <typename Tp>
struct Buf {
    Tp *p = nullptr; size_t count = 0;
    hipError_t alloc(size_t c) { release(); count = c; return c ? hipMalloc((void **)&p, c * sizeof(Tp)) : hipSuccess; }
    void release() { if (p) (void)hipFree(p); p = nullptr; count = 0; }
};
constexpr int ITERS_PER_GRAPH = 4;
static_assert(ITERS_PER_GRAPH % 2 == 0, "an iteration is an odd number of launches: an even count keeps the ping-pong parity");
constexpr int SEG_KMAX_LIMIT = 24;
}  // namespace

struct SegSolver {
    int print_info = 0, device = 0;
    int n = 0, nnz = 0, rows = 0, cols = 0;
    double c = 0.0;
    std::vector<int> rowptr, colidx;
    std::vector<double> vals, orgb;
    std::vector<int> left_idx, xi_left_idx;   // original indices of the live variables (ascending)
    int xi_rows = 0;
    bool has_problem = false, uploaded = false, inited = false, xi_valid = false;
    int G = 0, EPT = 2, kmax = 10, parity = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // one instantiated graph (ITERS_PER_GRAPH iterations) per launch count kmax and start parity
    hipGraph_t graph[SEG_KMAX_LIMIT + 1][2] = {}; hipGraphExec_t gexec[SEG_KMAX_LIMIT + 1][2] = {};
    bool adaptive = true;
    double kernel_ms = 0.0; long long launches = 0;
    Buf<int> d_ecol, d_left;
    Buf<uint8_t> d_rowlen;
    int ell_w = 0;
    bool dia = false; int doff[7] = {0, 0, 0, 0, 0, 0, 0};      // diagonal storage of the matrix (SegDev::dia), chosen by upload()
    Buf<unsigned long long> d_dpack; Buf<double> d_adiag;
    Buf<double> d_eval, x, y1, y2, z1, z2, b, rhs, r, z, tmp, dinv, td, p0, p1, part, xhist, xi_out;
    Buf<uint8_t> live, fixval, newfix;
    Buf<SegState> st;
    int ws_cap = 0, last_ws = 0;
    int record = 0;        // legacy loop keeps x of up to `record` iterations (lpbox_set_record; print_info 1, SEGcpp:1209-1213,1270-1277)
    int rec_cols = 0;      // iterations the last legacy solve recorded
    SegState hst;

    SegDev dev() const {
        SegDev d;
        d.n = n; d.nnz = nnz; d.G = G; d.EPT = EPT;
        d.ecol = d_ecol.p; d.eval = d_eval.p; d.rowlen = d_rowlen.p; d.ell_w = ell_w;
        d.dia = dia ? 1 : 0; d.dpack = d_dpack.p; d.adiag = d_adiag.p;
        for (int k = 0; k < 7; k++) d.doff[k] = doff[k];
        d.x = x.p; d.y1 = y1.p; d.y2 = y2.p; d.z1 = z1.p; d.z2 = z2.p; d.b = b.p; d.rhs = rhs.p; d.r = r.p; d.z = z.p;
        d.tmp = tmp.p; d.dinv = dinv.p; d.td = td.p; d.p0 = p0.p; d.p1 = p1.p; d.live = live.p; d.fixval = fixval.p;
        d.newfix = newfix.p; d.part = part.p; d.xhist = xhist.p; d.ws_cap = ws_cap; d.st = st.p;
        d.c1_init = std::pow((double)n, 1.0 / 2);      // pow(n, 1/p), p = 2 (SEGcpp:557,670)
        return d;
    }
};

namespace {

int use_device(SegSolver *s) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return lpbox_fail(LPBOX_E_NODEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(s->device));
    return LPBOX_OK;
}

// cv::resize(src, dst, Size(), scale, scale) with INTER_LINEAR on 8-bit data (SEGcpp:705-714), following OpenCV 4.4's
// fixed-point scheme: 11-bit coefficients, horizontal pass in int, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
void resize_linear_u8(const unsigned char *src, int rows, int cols, double scale, std::vector<unsigned char> &dst, int &orows, int &ocols) {
    orows = (int)std::lround(rows * scale); ocols = (int)std::lround(cols * scale);
    dst.assign((size_t)orows * ocols, 0);
    const double inv = 1.0 / scale;
    std::vector<int> xo(ocols), yo(orows);
    std::vector<short> ax(2 * (size_t)ocols), ay(2 * (size_t)orows);
    auto coef = [&](int dpos, int slen, int &ofs, short &c0, short &c1) {
        float f = (float)((dpos + 0.5) * inv - 0.5);
        int sp = (int)std::floor(f); f -= sp;
        if (sp < 0) { f = 0; sp = 0; }
        if (sp >= slen - 1) { f = 0; sp = slen - 1; }
        ofs = sp; c0 = (short)std::lrintf((1.f - f) * 2048); c1 = (short)std::lrintf(f * 2048);
    };
    for (int dx = 0; dx < ocols; dx++) coef(dx, cols, xo[dx], ax[2 * dx], ax[2 * dx + 1]);
    for (int dy = 0; dy < orows; dy++) coef(dy, rows, yo[dy], ay[2 * dy], ay[2 * dy + 1]);
    for (int dy = 0; dy < orows; dy++) {
        const unsigned char *s0 = src + (size_t)yo[dy] * cols, *s1 = src + (size_t)std::min(yo[dy] + 1, rows - 1) * cols;
        for (int dx = 0; dx < ocols; dx++) {
            const int x0 = xo[dx], x1 = std::min(x0 + 1, cols - 1);
            const int r0 = s0[x0] * ax[2 * dx] + s0[x1] * ax[2 * dx + 1], r1 = s1[x0] * ax[2 * dx] + s1[x1] * ax[2 * dx + 1];
            const int v = (((ay[2 * dy] * (r0 >> 4)) >> 16) + ((ay[2 * dy + 1] * (r1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * ocols + dx] = (unsigned char)std::clamp(v, 0, 255);
        }
    }
}

// sum in the association of Eigen's vectorised redux (two 2-wide accumulators; Core/Redux.h), used by .mean()/.sum()
double eigen_sum(const std::vector<double> &a) {
    const size_t n = a.size();
    if (n == 0) return 0.0;
    const size_t e2 = (n / 4) * 4, e1 = (n / 2) * 2;
    if (e1 == 0) { double r = a[0]; for (size_t i = 1; i < n; i++) r = r + a[i]; return r; }
    double p0a = a[0], p0b = a[1];
    if (e1 > 2) {
        double p1a = a[2], p1b = a[3];
        for (size_t i = 4; i < e2; i += 4) { p0a += a[i]; p0b += a[i + 1]; p1a += a[i + 2]; p1b += a[i + 3]; }
        p0a = p0a + p1a; p0b = p0b + p1b;
        if (e1 > e2) { p0a = p0a + a[e2]; p0b = p0b + a[e2 + 1]; }
    }
    double r = p0a + p0b;
    for (size_t i = e1; i < n; i++) r = r + a[i];
    return r;
}

// SEGcpp:46-248 (+ :727 scale by 1/263, :747 rounding of the unary costs, :755-756 A_ptr = A/2)
void build_costs(const unsigned char *gray, int rows, int cols, std::vector<int> &rowptr, std::vector<int> &colidx,
                 std::vector<double> &vals, std::vector<double> &b, double &c) {
    const int n = rows * cols;
    std::vector<double> v(n);                                   // vectorize(): column-major flatten (:46-53)
    for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) v[(size_t)j * rows + i] = gray[(size_t)i * cols + j] / 263.0;
    const double sigma = 0.1, bb = 0.6, f1 = 0.2, f2 = 0.2;     // :736-739
    const double cc = std::log(2.0 * M_PI) / 2.0 + std::log(sigma);
    std::vector<double> U1(n);
    b.assign(n, 0.0);
    for (int p = 0; p < n; p++) {                               // get_unary_cost :55-81
        const double ab = std::pow(v[p] - bb, 2.0) / (2 * sigma * sigma) + cc;
        const double aa = std::exp(-std::pow(v[p] - f1, 2.0) / (2 * sigma * sigma)) + std::exp(-std::pow(v[p] - f2, 2) / (2 * sigma * sigma));
        const double af = -std::log(aa + 2.220446049250313e-16) + cc + std::log(2.0);
        U1[p] = std::round(ab);
        b[p] = std::round(af) - U1[p];                          // b = U2 - U1 (:232)
    }
    c = eigen_sum(U1);                                          // :245
    const double mean = eigen_sum(v) / n;                       // get_binary_cost :173-224
    std::vector<double> sq(n);
    for (int p = 0; p < n; p++) sq[p] = (v[p] - mean) * (v[p] - mean);
    const double sig = std::sqrt(eigen_sum(sq) / (n - 1));      // sample std, used where a variance is usual (Q3)
    rowptr.assign((size_t)n + 1, 0); colidx.clear(); vals.clear();
    colidx.reserve((size_t)7 * n); vals.reserve((size_t)7 * n);
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
        const int r = i * cols + j;                             // pairs are numbered row-major (:157-158) ...
        rowptr[r] = (int)colidx.size();
        const size_t first = colidx.size();
        for (int a = -1; a <= 1; a++) for (int bq = -1; bq <= 1; bq++) {
            if (a == 0 && bq == 0) { colidx.push_back(r); vals.push_back(0.0); continue; }   // explicit zero diagonal (:213-219)
            if (a == bq || i + a < 0 || i + a >= rows || j + bq < 0 || j + bq >= cols) continue;   // a != b: 6 neighbours (:153-155)
            const int q = (i + a) * cols + (j + bq);
            const double dI = std::pow(v[r] - v[q], 2.0) / sig; // ... but intensities are read column-major (:192-193)
            colidx.push_back(q); vals.push_back(std::round(3 * std::exp(-dI)));
        }
        double We = 0;                                          // get_A_b_from_cost :226-248
        for (size_t e = first; e < colidx.size(); e++) We += (-vals[e]) * 1.0;
        We = -(0.0 + 1.0 * We);
        for (size_t e = first; e < colidx.size(); e++) {
            double a_e = -vals[e];
            if (colidx[e] == r) a_e += We;
            vals[e] = (2 * a_e) / 2;
        }
    }
    rowptr[n] = (int)colidx.size();
}

// Can the matrix be held as diagonals (SegDev::dia)?  At most three distinct column offsets either side of the main diagonal, a stored
// diagonal entry in every row, every off-diagonal value equal to -w for an integer w in 0..255.  True for every problem the image cost
// builder makes (SEGcpp:144-248: offsets {+-1, +-(ncols-1), +-ncols}, w = round(3 exp(.))); anything else keeps the ELL form.
bool as_diagonals(const SegSolver *s, int (&doff)[7], std::vector<unsigned long long> &dpack, std::vector<double> &adiag) {
    const int n = s->n;
    std::vector<int> neg, pos;
    auto note = [](std::vector<int> &v, int off) {
        for (int o : v) if (o == off) return true;
        if (v.size() == 3) return false;
        v.push_back(off);
        return true;
    };
    for (int i = 0; i < n; i++) {
        bool have_diag = false;
        for (int k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) {
            const int off = s->colidx[k] - i;
            if (off == 0) { have_diag = true; continue; }
            const double v = s->vals[k];
            if (!(v <= 0.0 && v >= -255.0 && v == std::floor(v))) return false;
            if (!note(off < 0 ? neg : pos, off)) return false;
        }
        if (!have_diag) return false;
    }
    std::sort(neg.begin(), neg.end()); std::sort(pos.begin(), pos.end());
    // slots 0..2: negative offsets ascending (missing ones in front), 3: the diagonal, 4..6: positive ascending (missing ones behind);
    // a missing slot points at a harmless neighbour and carries w = 0 in every row
    for (int k = 0; k < 3; k++) doff[k] = -1, doff[4 + k] = 1;
    doff[3] = 0;
    std::vector<std::pair<int, int>> slot_of;                     // (offset, byte of dpack)
    for (size_t k = 0; k < neg.size(); k++) { const int q = 3 - (int)neg.size() + (int)k; doff[q] = neg[k]; slot_of.push_back({neg[k], q}); }
    for (size_t k = 0; k < pos.size(); k++) { const int q = 4 + (int)k; doff[q] = pos[k]; slot_of.push_back({pos[k], q - 1}); }
    dpack.assign((size_t)n, 0ull); adiag.assign((size_t)n, 0.0);
    for (int i = 0; i < n; i++)
        for (int k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) {
            const int off = s->colidx[k] - i;
            if (off == 0) { adiag[i] = s->vals[k]; continue; }
            for (auto &so : slot_of)
                if (so.first == off) dpack[i] |= (unsigned long long)(unsigned)(-s->vals[k]) << (8 * so.second);
        }
    return true;
}

int upload(SegSolver *s) {
    int rc = use_device(s);
    if (rc) return rc;
    const int n = s->n;
    s->EPT = 2;
    if (const char *e = getenv("LPBOX_SEG_EPT")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) s->EPT = v; }   // tuning only
    while ((n + SEG_T * s->EPT - 1) / (SEG_T * s->EPT) > 2 * SEG_T && s->EPT < 64) s->EPT *= 2;   // keep G <= 512 partials
    s->G = (n + SEG_T * s->EPT - 1) / (SEG_T * s->EPT);
    if (s->G > 2 * SEG_T) return lpbox_fail(LPBOX_E_UNSUPPORTED, "n = %d is beyond the two-level reduction of the segmentation kernels", n);
    if (!s->stream) HIPCHK(hipStreamCreate(&s->stream));
    if (!s->ev0) { HIPCHK(hipEventCreate(&s->ev0)); HIPCHK(hipEventCreate(&s->ev1)); }
    int w = 0;
    for (int i = 0; i < n; i++) w = std::max(w, s->rowptr[i + 1] - s->rowptr[i]);
    if (w > 255) return lpbox_fail(LPBOX_E_UNSUPPORTED, "a row of A stores %d entries; the ELL layout of the segmentation kernels holds at most 255", w);
    s->ell_w = w;
    std::vector<unsigned long long> dpack; std::vector<double> adiag;
    s->dia = n >= 2 && w <= 7 && !getenv("LPBOX_SEG_NODIA") && as_diagonals(s, s->doff, dpack, adiag);      // LPBOX_SEG_NODIA: keep ELL (A/B, tests)
    if (s->dia) {
        s->d_ecol.release(); s->d_eval.release();
        HIPCHK(s->d_dpack.alloc(n)); HIPCHK(s->d_adiag.alloc(n));
        HIPCHK(hipMemcpy(s->d_dpack.p, dpack.data(), sizeof(unsigned long long) * (size_t)n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->d_adiag.p, adiag.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    } else {
        s->d_dpack.release(); s->d_adiag.release();
        HIPCHK(s->d_ecol.alloc((size_t)w * n)); HIPCHK(s->d_eval.alloc((size_t)w * n));
    }
    HIPCHK(s->d_rowlen.alloc(n)); HIPCHK(s->d_left.alloc(n));
    for (Buf<double> *bp : {&s->x, &s->y1, &s->y2, &s->z1, &s->z2, &s->b, &s->rhs, &s->r, &s->z, &s->tmp, &s->dinv, &s->td, &s->p0, &s->p1})
        HIPCHK(bp->alloc(n));
    HIPCHK(s->live.alloc(n)); HIPCHK(s->fixval.alloc(n)); HIPCHK(s->newfix.alloc(n));
    HIPCHK(s->part.alloc((size_t)5 * SEG_NPART * s->G)); HIPCHK(s->st.alloc(2));
    if (!s->dia) {
        std::vector<int> ecol((size_t)w * n); std::vector<double> eval((size_t)w * n, 0.0); std::vector<uint8_t> rl(n);
        for (int i = 0; i < n; i++) {
            const int len = s->rowptr[i + 1] - s->rowptr[i];
            rl[i] = (uint8_t)len;
            for (int k = 0; k < w; k++) {
                ecol[(size_t)k * n + i] = k < len ? s->colidx[s->rowptr[i] + k] : i;
                if (k < len) eval[(size_t)k * n + i] = s->vals[s->rowptr[i] + k];
            }
        }
        HIPCHK(hipMemcpy(s->d_ecol.p, ecol.data(), sizeof(int) * ecol.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->d_eval.p, eval.data(), sizeof(double) * eval.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->d_rowlen.p, rl.data(), rl.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemset(s->part.p, 0, sizeof(double) * (size_t)5 * SEG_NPART * s->G));
    HIPCHK(hipMemset(s->newfix.p, 0, n));
    s->uploaded = true;
    return LPBOX_OK;
}

int read_state(SegSolver *s) {
    HIPCHK(hipMemcpyAsync(&s->hst, s->st.p + s->parity, sizeof(SegState), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return LPBOX_OK;
}

// one hipGraph = ITERS_PER_GRAPH (even) outer iterations of 3 + 2 kmax launches: an even number of launches, so the state ping-pong
// parity is preserved across replays
int ensure_graph(SegSolver *s) {
    if (s->gexec[s->kmax][s->parity]) return LPBOX_OK;
    int par = s->parity;
    HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = seg_enqueue_iterations(s->dev(), ITERS_PER_GRAPH, s->kmax, &par, s->stream);
    hipError_t e2 = hipStreamEndCapture(s->stream, &s->graph[s->kmax][s->parity]);
    if (e != hipSuccess || e2 != hipSuccess) return lpbox_fail(LPBOX_E_HIP, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    HIPCHK(hipGraphInstantiate(&s->gexec[s->kmax][s->parity], s->graph[s->kmax][s->parity], nullptr, nullptr, 0));
    return LPBOX_OK;
}

void drop_graphs(SegSolver *s) {
    for (int k = 0; k <= SEG_KMAX_LIMIT; k++) for (int p = 0; p < 2; p++) {
        if (s->gexec[k][p]) { (void)hipGraphExecDestroy(s->gexec[k][p]); s->gexec[k][p] = nullptr; }
        if (s->graph[k][p]) { (void)hipGraphDestroy(s->graph[k][p]); s->graph[k][p] = nullptr; }
    }
}

// run iterations up to iter_end (the window must already be set); returns when stopped / window done / all fixed.
// Each batch enqueues kmax (matvec, update) pairs per outer iteration; kernels fall through once the PCG has converged, so
// kmax only has to cover the iteration counts seen recently (pcg_max of the previous batch + 2); a batch that runs out of
// pairs halts itself (SEG_HALT_PCG_MORE) and is resumed here.
int run_window(SegSolver *s, int iter_end) {
    static const bool nograph = getenv("LPBOX_SEG_NOGRAPH") != nullptr;   // eager launches (profilers that dislike graphs)
    // spare (matvec, update) pairs per iteration beyond the largest PCG count of the previous batch of <= 32 iterations.  A spare pair
    // is a launch that falls through (~3 us); a PCG that needs more than was enqueued halts the chain and is resumed (a host round
    // trip).  Measured over 100 images at 10^4 nodes and the full-resolution one: 0 spare pairs 8 % / 7 % faster than 2, 1 in between
    // -- the counts drift slowly, misses are rare.  LPBOX_SEG_KMARGIN overrides (also for the multi-problem chain, which defaults to 2:
    // there one late problem stalls all).
    static const int kmargin = getenv("LPBOX_SEG_KMARGIN") ? std::max(0, atoi(getenv("LPBOX_SEG_KMARGIN"))) : 0;
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    for (;;) {
        int rc = read_state(s);
        if (rc) return rc;
        if (s->hst.halt == SEG_HALT_PCG_MORE) {
            HIPCHK(seg_enqueue_pcg_more(s->dev(), 16, &s->parity, s->stream));
            s->launches += 34;
            if (s->adaptive) s->kmax = std::min(SEG_KMAX_LIMIT, std::max(s->kmax, s->hst.pcg_k + 4));
            continue;
        }
        if (s->hst.halt != SEG_HALT_NONE) break;
        const int remaining = iter_end - s->hst.iter;
        if (remaining <= 0 && !s->hst.have_prev) break;
        if (s->adaptive && s->hst.outer_total > 0) s->kmax = std::min(SEG_KMAX_LIMIT, std::max(2, s->hst.pcg_max + kmargin));
        HIPCHK(seg_launch_copy(s->dev(), 1, &s->parity, s->stream));       // pcg_max = 0 for the coming batch
        HIPCHK(seg_enqueue_prep(s->dev(), &s->parity, s->stream));         // head of the batch (inside it post + yrhs do prep's work)
        s->launches += 2;
        const int per_launch = 3 + 2 * s->kmax;
        int batch = std::min(std::max(remaining, 0), 32);
        if (!nograph) { rc = ensure_graph(s); if (rc) return rc; }
        while (!nograph && batch >= ITERS_PER_GRAPH) {
            HIPCHK(hipGraphLaunch(s->gexec[s->kmax][s->parity], s->stream));
            batch -= ITERS_PER_GRAPH; s->launches += (long long)ITERS_PER_GRAPH * per_launch;
        }
        if (batch > 0) { HIPCHK(seg_enqueue_iterations(s->dev(), batch, s->kmax, &s->parity, s->stream)); s->launches += (long long)batch * per_launch; }
        HIPCHK(seg_enqueue_finalize(s->dev(), &s->parity, s->stream));
        s->launches += 1;
    }
    HIPCHK(hipEventRecord(s->ev1, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->kernel_ms += ms;
    return LPBOX_OK;
}

}  // namespace

SegSolver *segc_create(int print_info, int device) {
    SegSolver *s = new SegSolver();
    s->print_info = print_info; s->device = device;
    memset(&s->hst, 0, sizeof(s->hst));
    if (const char *e = getenv("LPBOX_SEG_KMAX")) { int v = atoi(e); if (v >= 1 && v <= SEG_KMAX_LIMIT) { s->kmax = v; s->adaptive = false; } }
    return s;
}

void segc_destroy(SegSolver *s) {
    if (!s) return;
    if (s->uploaded) (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    drop_graphs(s);
    s->d_ecol.release(); s->d_rowlen.release(); s->d_left.release(); s->d_eval.release(); s->d_dpack.release(); s->d_adiag.release();
    for (Buf<double> *bp : {&s->x, &s->y1, &s->y2, &s->z1, &s->z2, &s->b, &s->rhs, &s->r, &s->z, &s->tmp, &s->dinv, &s->td, &s->p0, &s->p1,
                            &s->part, &s->xhist, &s->xi_out})
        bp->release();
    s->live.release(); s->fixval.release(); s->newfix.release(); s->st.release();
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int segc_set_problem(SegSolver *s, int n, int nnz, const int *rowptr, const int *colidx, const double *vals, const double *b,
                     double c, int rows, int cols) {
    if (s->uploaded) return lpbox_fail(LPBOX_E_STATE, "problem already uploaded; create a new handle to change it");
    if (n <= 0 || nnz <= 0 || !rowptr || !colidx || !vals || !b) return lpbox_fail(LPBOX_E_BADARG, "bad problem arguments");
    if (rowptr[0] != 0 || rowptr[n] != nnz) return lpbox_fail(LPBOX_E_BADARG, "rowptr does not span nnz");
    for (int i = 0; i < n; i++) {
        bool diag = false;
        if (rowptr[i + 1] < rowptr[i]) return lpbox_fail(LPBOX_E_BADARG, "rowptr not monotone");
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++) {
            if (colidx[k] < 0 || colidx[k] >= n) return lpbox_fail(LPBOX_E_BADARG, "column index out of range");
            if (k > rowptr[i] && colidx[k] <= colidx[k - 1]) return lpbox_fail(LPBOX_E_BADARG, "column indices must ascend inside a row");
            diag |= colidx[k] == i;
        }
        if (!diag) return lpbox_fail(LPBOX_E_BADARG, "row %d stores no diagonal entry (the reference always stores one, SEGcpp:213-219)", i);
    }
    s->n = n; s->nnz = nnz; s->rows = rows; s->cols = cols; s->c = c;
    s->rowptr.assign(rowptr, rowptr + n + 1); s->colidx.assign(colidx, colidx + nnz); s->vals.assign(vals, vals + nnz);
    s->orgb.assign(b, b + n);
    s->has_problem = true;
    return LPBOX_OK;
}

int segc_set_image(SegSolver *s, const unsigned char *gray, int rows, int cols, int num_nodes) {
    if (!gray || rows <= 0 || cols <= 0 || num_nodes <= 0) return lpbox_fail(LPBOX_E_BADARG, "bad image arguments");
    const double scale = std::sqrt(num_nodes / (double)((long)rows * cols));     // SEGcpp:707
    std::vector<unsigned char> scaled;
    int sr = rows, sc = cols;
    const unsigned char *src = gray;
    if (scale != 1.0) { resize_linear_u8(gray, rows, cols, scale, scaled, sr, sc); src = scaled.data(); }
    if (sr < 2 || sc < 2) return lpbox_fail(LPBOX_E_BADARG, "scaled image %dx%d too small", sr, sc);
    std::vector<int> rowptr, colidx; std::vector<double> vals, b; double c = 0;
    build_costs(src, sr, sc, rowptr, colidx, vals, b, c);
    return segc_set_problem(s, sr * sc, (int)colidx.size(), rowptr.data(), colidx.data(), vals.data(), b.data(), c, sr, sc);
}

int segc_init(SegSolver *s) {
    if (!s->has_problem) return lpbox_fail(LPBOX_E_STATE, "no problem set");
    int rc = s->uploaded ? use_device(s) : upload(s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->b.p, s->orgb.data(), sizeof(double) * (size_t)s->n, hipMemcpyHostToDevice, s->stream));
    s->left_idx.resize(s->n);
    for (int i = 0; i < s->n; i++) s->left_idx[i] = i;
    s->xi_valid = false; s->parity = 0;
    HIPCHK(seg_launch_init(s->dev(), std::pow((double)s->n, 1.0 / 2), s->stream));    // pow(n, 1/p), p = 2 (SEGcpp:557,670)
    rc = read_state(s);
    if (rc) return rc;
    s->inited = true;
    return 1;
}

int segc_legacy(SegSolver *s, int *energy) {                                  // ADMM_bqp_unconstrained_legacy SEGcpp:1200-1380
    if (!s->inited) return lpbox_fail(LPBOX_E_STATE, "solve_init has not been called");
    int rc = use_device(s);
    if (rc) return rc;
    if (s->record > 0 && (!s->xhist.p || s->ws_cap < s->record)) {
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(s->xhist.alloc((size_t)s->record * s->n)); s->ws_cap = s->record; drop_graphs(s);
        HIPCHK(hipMemsetAsync(s->xhist.p, 0, sizeof(double) * (size_t)s->ws_cap * s->n, s->stream));
    }
    HIPCHK(seg_launch_set_window(s->dev(), 0, SEG_MAX_ITERS, s->record > 0 ? 2 : 0, &s->parity, s->stream));
    rc = run_window(s, SEG_MAX_ITERS);
    if (rc) return rc;
    s->rec_cols = s->record > 0 ? std::min(s->hst.cc, s->ws_cap) : 0;
    s->xi_valid = false;
    if (energy) *energy = (int)(s->hst.cur_obj + s->c);                        // :1379
    return LPBOX_OK;
}

// ADMM_bqp_unconstrained_init + _legacy (SEGcpp:658-810, 1200-1380) for B problems advanced in LOCKSTEP by one launch chain
// (image_segmentation.cpp:24-29 solves images 0..99 at 10^4 nodes one after the other: there a solve is launch-bound and the GPU
// mostly idle).  Every problem keeps its own control state, so the arithmetic of each is exactly that of a solve on its own.
int segc_legacy_batch(SegSolver **ss, int B, int *energies) {
    if (!ss || B <= 0) return lpbox_fail(LPBOX_E_BADARG, "empty batch");
    SegSolver *s0 = ss[0];
    for (int i = 0; i < B; i++) {
        if (!ss[i] || !ss[i]->has_problem) return lpbox_fail(LPBOX_E_STATE, "problem %d of the batch has no image / problem", i);
        if (ss[i]->device != s0->device) return lpbox_fail(LPBOX_E_BADARG, "all problems of a batch must live on one device");
        if (ss[i]->record > 0) return lpbox_fail(LPBOX_E_UNSUPPORTED, "recording is per-solver (lpbox_seg_legacy)");
        for (int k = 0; k < i; k++)                                 // one handle twice = every launch advancing the same buffers twice
            if (ss[k] == ss[i]) return lpbox_fail(LPBOX_E_BADARG, "problem %d and problem %d of the batch are the same handle", k, i);
    }
    // a solver counts as initialised only once the whole chain has completed: an early error return leaves every member of the
    // batch in the "solve_init has not been called" state instead of flagged ready with half-written device state
    for (int i = 0; i < B; i++) { ss[i]->inited = false; ss[i]->xi_valid = false; }
    int rc = LPBOX_OK;
    int Gmax = 0;
    for (int i = 0; i < B; i++) {                                   // upload + host-side reset of segc_init (the init KERNEL runs batched)
        SegSolver *s = ss[i];
        rc = s->uploaded ? use_device(s) : upload(s);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(s->b.p, s->orgb.data(), sizeof(double) * (size_t)s->n, hipMemcpyHostToDevice, s->stream));
        s->left_idx.resize(s->n);
        for (int k = 0; k < s->n; k++) s->left_idx[k] = k;
        Gmax = std::max(Gmax, s->G);
    }
    for (int i = 0; i < B; i++) HIPCHK(hipStreamSynchronize(ss[i]->stream));
    hipStream_t st = s0->stream;
    std::vector<SegDev> hd(B);
    for (int i = 0; i < B; i++) hd[i] = ss[i]->dev();
    Buf<SegDev> devs; Buf<SegState> dstates;
    struct Release { Buf<SegDev> &a; Buf<SegState> &b; ~Release() { a.release(); b.release(); } } release_on_exit{devs, dstates};
    HIPCHK(devs.alloc(B)); HIPCHK(dstates.alloc(B));
    HIPCHK(hipMemcpyAsync(devs.p, hd.data(), sizeof(SegDev) * (size_t)B, hipMemcpyHostToDevice, st));
    std::vector<SegState> hs(B);
    int parity = 0, kmax = s0->kmax;
    long long launches = 0;
    auto read_states = [&]() -> int {
        HIPCHK(segb_collect_states(devs.p, B, parity, dstates.p, st));
        HIPCHK(hipMemcpyAsync(hs.data(), dstates.p, sizeof(SegState) * (size_t)B, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return LPBOX_OK;
    };
    // spare (matvec, update) pairs per iteration beyond the largest PCG count any problem showed in the previous batch of iterations: 0, as
    // in the single-problem chain (a miss halts that problem's PCG and the chain resumes it).  Measured, 100 problems at 10^4 nodes, fastest
    // / median of five runs: margin 2 119 / 120 ms, margin 1 111 / 112 ms, margin 0 108 / 108 ms; 16 problems 46 -> 41 ms.
    static const int bmargin = getenv("LPBOX_SEG_KMARGIN") ? std::max(0, atoi(getenv("LPBOX_SEG_KMARGIN"))) : 0;
    HIPCHK(hipEventRecord(s0->ev0, st));
    HIPCHK(segb_launch_init(devs.p, B, Gmax, st));
    HIPCHK(segb_launch_set_window(devs.p, B, 0, SEG_MAX_ITERS, 0, &parity, st));
    launches += 2;
    for (;;) {
        rc = read_states();
        if (rc) return rc;
        bool more = false, any_run = false;
        int remaining = 0, pcg_max = 0, pcg_k = 0;
        for (int i = 0; i < B; i++) {
            const SegState &h = hs[i];
            if (h.halt == SEG_HALT_PCG_MORE) { more = true; pcg_k = std::max(pcg_k, h.pcg_k); continue; }
            if (h.halt != SEG_HALT_NONE) continue;
            const int rem = SEG_MAX_ITERS - h.iter;
            if (rem <= 0 && !h.have_prev) continue;
            any_run = true; remaining = std::max(remaining, rem); pcg_max = std::max(pcg_max, h.outer_total > 0 ? h.pcg_max : kmax - bmargin);
        }
        if (more) {                                                 // some PCG ran out of launches: resume it (the others fall through)
            HIPCHK(segb_enqueue_pcg_more(devs.p, B, Gmax, 16, &parity, st));
            launches += 34;
            if (s0->adaptive) kmax = std::min(SEG_KMAX_LIMIT, std::max(kmax, pcg_k + 4));
            continue;
        }
        if (!any_run) break;
        if (s0->adaptive) kmax = std::min(SEG_KMAX_LIMIT, std::max(2, pcg_max + bmargin));
        HIPCHK(segb_launch_copy(devs.p, B, 1, &parity, st));
        const int batch = std::min(std::max(remaining, 0), 32);
        if (batch > 0) HIPCHK(segb_enqueue_iterations(devs.p, B, Gmax, batch, kmax, &parity, st));
        HIPCHK(segb_enqueue_finalize(devs.p, B, Gmax, &parity, st));
        launches += 2 + (long long)batch * (4 + 2 * kmax);
    }
    HIPCHK(hipEventRecord(s0->ev1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, s0->ev0, s0->ev1));
    for (int i = 0; i < B; i++) {
        SegSolver *s = ss[i];
        s->hst = hs[i]; s->parity = parity; s->rec_cols = 0; s->xi_valid = false; s->inited = true;
        if (energies) energies[i] = (int)(s->hst.cur_obj + s->c);     // :1379
    }
    s0->kernel_ms += ms; s0->launches += launches;
    return LPBOX_OK;
}

int segc_l2f(SegSolver *s, int iter_start, int iter_end, const double *vec, int num, int *ret) {   // SEGcpp:917-1195
    if (!s->inited) return lpbox_fail(LPBOX_E_STATE, "solve_init has not been called");
    const int ws = iter_end - iter_start;
    if (ws > SEG_XITERS_COLS) return lpbox_fail(LPBOX_E_BADARG, "window of %d iterations exceeds the %d columns of x_iters (SEGcpp:924)", ws, SEG_XITERS_COLS);
    int rc = use_device(s);
    if (rc) return rc;
    const int n_live = (int)s->left_idx.size();
    if (num < 0 || num > n_live) return lpbox_fail(LPBOX_E_BADARG, "fix count %d outside [0,%d]", num, n_live);
    if (num != 0) {
        if (!vec) return lpbox_fail(LPBOX_E_BADARG, "fix vector missing");
        int cnt = 0;
        for (int q = 0; q < n_live; q++) if (vec[q] == 1 || vec[q] == 0) cnt++;
        if (cnt != num) return lpbox_fail(LPBOX_E_BADARG, "vec fixes %d variables but num = %d", cnt, num);
    }
    HIPCHK(seg_launch_set_window(s->dev(), iter_start, iter_end, 3, &s->parity, s->stream));
    if (num != 0) {
        std::vector<uint8_t> nf(s->n, 0);
        std::vector<int> keep; keep.reserve(n_live - num);
        for (int q = 0; q < n_live; q++) {
            const int org = s->left_idx[q];
            if (vec[q] == 1) nf[org] = 2; else if (vec[q] == 0) nf[org] = 1; else keep.push_back(org);
        }
        s->left_idx.swap(keep);
        HIPCHK(hipMemcpy(s->newfix.p, nf.data(), nf.size(), hipMemcpyHostToDevice));
        HIPCHK(seg_launch_fix(s->dev(), n_live - num, std::pow((double)(n_live - num), 1.0 / 2), &s->parity, s->stream));
        HIPCHK(hipMemsetAsync(s->newfix.p, 0, s->n, s->stream));
    }
    s->xi_rows = n_live - num; s->xi_left_idx = s->left_idx;
    if (ws > 0 && (!s->xhist.p || s->ws_cap < ws)) {
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(s->xhist.alloc((size_t)SEG_XITERS_COLS * s->n)); s->ws_cap = SEG_XITERS_COLS; drop_graphs(s);
    }
    if (s->ws_cap > 0 && s->xhist.p) HIPCHK(hipMemsetAsync(s->xhist.p, 0, sizeof(double) * (size_t)s->ws_cap * s->n, s->stream));   // x_iters = Zero (SEGcpp:924): ALL staged columns, also those of an earlier, longer window
    s->rec_cols = 0;
    if (!s->left_idx.empty()) HIPCHK(hipMemcpyAsync(s->d_left.p, s->left_idx.data(), sizeof(int) * s->left_idx.size(), hipMemcpyHostToDevice, s->stream));
    rc = run_window(s, iter_end);
    if (rc) return rc;
    s->last_ws = ws; s->xi_valid = true;
    if (ret) *ret = s->hst.ret;
    return LPBOX_OK;
}

int segc_get_n(SegSolver *s) { return s->inited ? s->hst.n_live : s->n; }
int segc_get_org_n(SegSolver *s) { return s->n; }
int segc_get_iter(SegSolver *s) { return s->inited ? s->hst.iter : 0; }
int segc_get_shape(SegSolver *s, int *rows, int *cols) { if (rows) *rows = s->rows; if (cols) *cols = s->cols; return LPBOX_OK; }

int segc_set_record(SegSolver *s, int on) {
    s->record = on <= 0 ? 0 : (on == 1 ? SEG_REC_COLS : on);
    return LPBOX_OK;
}

int segc_get_x_history(SegSolver *s, int first, int count, double *out) {     // rows of ../xiter/<problem>.csv (SEGcpp:1270-1277)
    if (!out) return s->rec_cols;
    if (first < 0 || count < 0 || first + count > s->rec_cols)
        return lpbox_fail(LPBOX_E_BADARG, "iterations [%d,%d) outside the %d recorded", first, first + count, s->rec_cols);
    int rc = use_device(s);
    if (rc) return rc;
    if (count) HIPCHK(hipMemcpy(out, s->xhist.p + (size_t)first * s->n, sizeof(double) * (size_t)count * s->n, hipMemcpyDeviceToHost));
    return count;
}

int segc_get_x_iters(SegSolver *s, int ws, double *out) {                     // get_x_iters_d SEGcpp:839-851
    if (!s->xi_valid) return lpbox_fail(LPBOX_E_STATE, "solve_iter_l2f has not been called");
    if (ws < 0 || ws > SEG_XITERS_COLS) return lpbox_fail(LPBOX_E_BADARG, "ws = %d outside [0,%d]", ws, SEG_XITERS_COLS);
    const int rows = s->xi_rows;
    if (!out || rows == 0 || ws == 0) return rows;
    int rc = use_device(s);
    if (rc) return rc;
    if (s->xi_out.count < (size_t)rows * ws) HIPCHK(s->xi_out.alloc((size_t)s->n * SEG_XITERS_COLS));
    HIPCHK(seg_launch_pack_xiters(s->dev(), s->d_left.p, rows, ws, s->xi_out.p, s->stream));
    HIPCHK(hipMemcpyAsync(out, s->xi_out.p, sizeof(double) * (size_t)rows * ws, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return rows;
}

// the same (rows x ws) row-major window left on the device (stride = rows * ws doubles: one problem per handle)
int segc_get_x_iters_device(SegSolver *s, int ws, void **dev_ptr, long *stride) {
    if (!s->xi_valid) return lpbox_fail(LPBOX_E_STATE, "solve_iter_l2f has not been called");
    if (ws <= 0 || ws > SEG_XITERS_COLS) return lpbox_fail(LPBOX_E_BADARG, "ws = %d outside (0,%d]", ws, SEG_XITERS_COLS);
    int rc = use_device(s);
    if (rc) return rc;
    const int rows = s->xi_rows;
    if (s->xi_out.count < (size_t)std::max(rows, 1) * ws) HIPCHK(s->xi_out.alloc((size_t)s->n * SEG_XITERS_COLS));
    HIPCHK(seg_launch_pack_xiters(s->dev(), s->d_left.p, rows, ws, s->xi_out.p, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (dev_ptr) *dev_ptr = s->xi_out.p;
    if (stride) *stride = (long)rows * ws;
    return LPBOX_OK;
}

static int fetch_solution(SegSolver *s, std::vector<double> &sol) {            // get_x_sol SEGcpp:895-914
    std::vector<double> x(s->n); std::vector<uint8_t> live(s->n), fv(s->n);
    HIPCHK(hipMemcpy(x.data(), s->x.p, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(live.data(), s->live.p, s->n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fv.data(), s->fixval.p, s->n, hipMemcpyDeviceToHost));
    sol.resize(s->n);
    for (int i = 0; i < s->n; i++) sol[i] = live[i] ? (x[i] >= 0.5 ? 1.0 : 0.0) : (double)fv[i];
    return LPBOX_OK;
}

int segc_get_x_sol(SegSolver *s, double *out) {
    if (!s->inited || !out) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    int rc = use_device(s);
    if (rc) return rc;
    std::vector<double> sol;
    if ((rc = fetch_solution(s, sol))) return rc;
    memcpy(out, sol.data(), sizeof(double) * (size_t)s->n);
    return s->n;
}

int segc_get_obj(SegSolver *s, double *out) {                                 // get_final_obj SEGcpp:868-893
    if (!s->inited || !out) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    int rc = use_device(s);
    if (rc) return rc;
    std::vector<double> xx;
    if ((rc = fetch_solution(s, xx))) return rc;
    // compute_cost(xx, org_A, org_b) = xx.(A xx) + b.xx; every term is an integer, so the summation order is immaterial
    double val = 0.0, val2 = 0.0;
    for (int i = 0; i < s->n; i++) {
        double t = 0;
        for (int k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) t += s->vals[k] * xx[s->colidx[k]];
        val += xx[i] * (0.0 + 1.0 * t);
        val2 += s->orgb[i] * xx[i];
    }
    *out = (val + val2) + s->c;
    return LPBOX_OK;
}

int segc_get_config(SegSolver *s, int *threads, int *ept, int *groups) {
    if (!s->uploaded) { int rc = upload(s); if (rc) return rc; }
    if (threads) *threads = SEG_T;
    if (ept) *ept = s->EPT;
    if (groups) *groups = s->G;
    return LPBOX_OK;
}

int segc_get_counters(SegSolver *s, long long *outer, long long *pcg) {
    if (outer) *outer = s->hst.outer_total;
    if (pcg) *pcg = s->hst.pcg_total;
    return LPBOX_OK;
}

int segc_get_stop(SegSolver *s, int *reason, int *legacy_iter_p1) {
    if (reason) *reason = s->hst.stop;
    if (legacy_iter_p1) *legacy_iter_p1 = s->hst.legacy_iter_p1;
    return LPBOX_OK;
}

int segc_kernel_time(SegSolver *s, double *ms, long long *launches, int reset) {
    if (ms) *ms = s->kernel_ms;
    if (launches) *launches = s->launches;
    if (reset) { s->kernel_ms = 0.0; s->launches = 0; }
    return LPBOX_OK;
}

int segc_debug_vec(SegSolver *s, const char *name, double *out, int cap) {
    if (!s->inited) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    int rc = use_device(s);
    if (rc) return rc;
    const double *src = nullptr;
    if (!strcmp(name, "x")) src = s->x.p; else if (!strcmp(name, "z1")) src = s->z1.p; else if (!strcmp(name, "z2")) src = s->z2.p;
    else if (!strcmp(name, "b")) src = s->b.p; else if (!strcmp(name, "y1")) src = s->y1.p; else if (!strcmp(name, "y2")) src = s->y2.p;
    else if (!strcmp(name, "td")) src = s->td.p;
    else return lpbox_fail(LPBOX_E_BADARG, "unknown vector '%s'", name);
    if (cap < s->n) return lpbox_fail(LPBOX_E_BADARG, "buffer too small");
    HIPCHK(hipMemcpy(out, src, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost));
    return s->n;
}

int segc_debug_scalar(SegSolver *s, const char *name, double *out) {
    const SegState &h = s->hst;
    struct { const char *n; double v; } tab[] = {
        {"rho1", h.rho1}, {"gamma", h.gamma_val}, {"cur_obj", h.cur_obj}, {"std_obj", h.std_obj}, {"cvg1", h.cvg1}, {"cvg2", h.cvg2},
        {"obj_val", h.obj_val}, {"best_bin_obj", h.best_bin_obj}, {"c", s->c}, {"last_pcg", (double)h.last_pcg}, {"iter", (double)h.iter},
        {"matrix_as_diagonals", s->dia ? 1.0 : 0.0},
    };
    for (auto &e : tab) if (!strcmp(e.n, name)) { *out = e.v; return LPBOX_OK; }
    return lpbox_fail(LPBOX_E_BADARG, "unknown scalar '%s'", name);
}

int segc_get_problem(SegSolver *s, int *n, int *nnz, int *rowptr, int *colidx, double *vals, double *b, double *c) {
    if (!s->has_problem) return lpbox_fail(LPBOX_E_STATE, "no problem set");
    if (n) *n = s->n;
    if (nnz) *nnz = s->nnz;
    if (rowptr) memcpy(rowptr, s->rowptr.data(), sizeof(int) * ((size_t)s->n + 1));
    if (colidx) memcpy(colidx, s->colidx.data(), sizeof(int) * (size_t)s->nnz);
    if (vals) memcpy(vals, s->vals.data(), sizeof(double) * (size_t)s->nnz);
    if (b) memcpy(b, s->orgb.data(), sizeof(double) * (size_t)s->n);
    if (c) *c = s->c;
    return LPBOX_OK;
}
