// lpbox_gen_capi.hip -- host side + C-ABI (lpbox_bqp_* in include/lpbox_hip.h) of the GENERIC constrained binary-QP path:
// the reference's ADMM_bqp (SEGcpp:1384-1832) behind its four wrappers ADMM_bqp_unconstrained / _linear_eq / _linear_ineq /
// _linear_eq_and_uneq (SEGcpp:1834-2109).  The kernels are in lpbox_gen_kernels.hip.
#include "../../include/lpbox_hip.h"
#include "lpbox_gen.h"
#include "lpbox_capi_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return lpbox_fail(LPBOX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define CHK(expr) do { int rc_ = (expr); if (rc_ < 0) return rc_; } while (0)

namespace {
template <typename Tp>
struct Buf {
    Tp *p = nullptr; size_t count = 0;
    hipError_t alloc(size_t c) { release(); count = c; return hipMalloc((void **)&p, std::max<size_t>(c, 1) * sizeof(Tp)); }
    void release() { if (p) (void)hipFree(p); p = nullptr; count = 0; }
    hipError_t upload(const std::vector<Tp> &v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(Tp), hipMemcpyHostToDevice);
    }
};
struct HostCsr { int rows = 0, cols = 0; std::vector<int> ptr, idx; std::vector<double> val; };
struct DevCsr { Buf<int> ptr, idx; Buf<double> val; GenCsr view() const { return GenCsr{ptr.p, idx.p, val.p}; } };

int load_csr(HostCsr &M, int rows, int cols, const int *ptr, const int *idx, const double *val, const char *what) {
    if (!ptr || ptr[0] != 0) return lpbox_fail(LPBOX_E_BADARG, "%s: bad row pointer", what);
    for (int i = 0; i < rows; i++) {
        if (ptr[i + 1] < ptr[i]) return lpbox_fail(LPBOX_E_BADARG, "%s: row pointer not monotone", what);
        for (int k = ptr[i]; k < ptr[i + 1]; k++) {
            if (idx[k] < 0 || idx[k] >= cols) return lpbox_fail(LPBOX_E_BADARG, "%s: column index out of range", what);
            if (k > ptr[i] && idx[k] <= idx[k - 1]) return lpbox_fail(LPBOX_E_BADARG, "%s: columns must ascend inside a row", what);
        }
    }
    M.rows = rows; M.cols = cols;
    M.ptr.assign(ptr, ptr + rows + 1); M.idx.assign(idx, idx + ptr[rows]); M.val.assign(val, val + ptr[rows]);
    return LPBOX_OK;
}
void transpose(const HostCsr &S, HostCsr &Tt) {        // rows of the result = columns of S, entries in ascending original row order
    Tt.rows = S.cols; Tt.cols = S.rows;
    Tt.ptr.assign((size_t)S.cols + 1, 0);
    for (int c : S.idx) Tt.ptr[c + 1]++;
    for (int j = 0; j < S.cols; j++) Tt.ptr[j + 1] += Tt.ptr[j];
    Tt.idx.resize(S.idx.size()); Tt.val.resize(S.val.size());
    std::vector<int> cur(Tt.ptr.begin(), Tt.ptr.end() - 1);
    for (int i = 0; i < S.rows; i++)
        for (int k = S.ptr[i]; k < S.ptr[i + 1]; k++) { const int p = cur[S.idx[k]]++; Tt.idx[p] = i; Tt.val[p] = S.val[k]; }
}
hipError_t upload(DevCsr &D, const HostCsr &H) {
    hipError_t e = D.ptr.upload(H.ptr); if (e != hipSuccess) return e;
    e = D.idx.upload(H.idx); if (e != hipSuccess) return e;
    return D.val.upload(H.val);
}
}  // namespace

struct lpbox_bqp {
    int device = 0, n = 0, m = 0, l = 0;
    GenParams prm;
    HostCsr A, C, Ct, E, Et;
    std::vector<int> adiag;
    std::vector<double> b, x0, d, f;
    bool has_problem = false, uploaded = false, solved = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int G = 0, Gm = 0, Gl = 0, EPT = 2, EPTm = 2, EPTl = 2, kmax = 12, parity = 0;
    bool adaptive = true;
    double kernel_ms = 0.0; long long launches = 0;
    // launch-bound inner loop: GEN_ITERS_PER_GRAPH outer iterations captured once per (PCG launch count, start parity) and replayed
    std::map<std::pair<int, int>, hipGraphExec_t> gexec; std::vector<hipGraph_t> graphs;
    long long graph_launches = 0;
    bool use_graph = true;
    DevCsr dA, dCr, dCc, dEr, dEc;
    Buf<int> d_adiag;
    Buf<double> tmval, Cc_sv, Ec_sv, x, xt, y1, y2, z1, z2, db, rhs, r, z, tmp, p0, p1, gsrc, pdiag, dinv, Csq, Esq, best, dx0,
        z3, qC, dd, y3, z4, fy, qE, Ex, df, part, red;
    Buf<double2> zp;
    Buf<GenState> st;
    GenState hst;

    GenDev dev() const {
        GenDev g;
        g.n = n; g.m = m; g.l = l; g.G = G; g.Gm = Gm; g.Gl = Gl; g.EPT = EPT; g.EPTm = EPTm; g.EPTl = EPTl;
        g.eq = m > 0; g.ineq = l > 0; g.prm = prm;
        g.aptr = dA.ptr.p; g.aidx = dA.idx.p; g.aval = dA.val.p; g.tmval = tmval.p; g.adiag = d_adiag.p;
        g.Cr = dCr.view(); g.Cc = dCc.view(); g.Er = dEr.view(); g.Ec = dEc.view(); g.Cc_sv = Cc_sv.p; g.Ec_sv = Ec_sv.p;
        g.Cnnz = (int)C.idx.size(); g.Ennz = (int)E.idx.size();
        g.x = x.p; g.xt = xt.p; g.y1 = y1.p; g.y2 = y2.p; g.z1 = z1.p; g.z2 = z2.p; g.rhs = rhs.p; g.r = r.p; g.z = z.p; g.tmp = tmp.p;
        g.p0 = p0.p; g.p1 = p1.p; g.gsrc = gsrc.p; g.pdiag = pdiag.p; g.dinv = dinv.p; g.Csq = Csq.p; g.Esq = Esq.p; g.best = best.p;
        g.b = db.p; g.zp = zp.p; g.z3 = z3.p; g.qC = qC.p; g.d = dd.p; g.y3 = y3.p; g.z4 = z4.p; g.fy = fy.p; g.qE = qE.p; g.Ex = Ex.p;
        g.f = df.p; g.part = part.p; g.red = red.p; g.st = st.p;
        return g;
    }
};

namespace {
int use_device(lpbox_bqp *h) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return lpbox_fail(LPBOX_E_NODEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(h->device));
    return LPBOX_OK;
}
#define FIN(nv) do { HIPCHK(gen_launch_fin(d, nv, h->stream)); h->launches++; } while (0)
#define ROWS(mode) do { HIPCHK(gen_launch_rows(d, mode, &h->parity, h->stream)); h->launches++; } while (0)

int enqueue_pcg(lpbox_bqp *h, const GenDev &d, int pairs) {
    for (int k = 0; k < pairs; k++) {
        ROWS(1);
        HIPCHK(gen_launch_pcg_cols(d, &h->parity, h->stream)); h->launches++;
        FIN(1);
        HIPCHK(gen_launch_pcg_upd(d, &h->parity, h->stream)); h->launches++;
        FIN(2);
    }
    return LPBOX_OK;
}
int enqueue_tail(lpbox_bqp *h, const GenDev &d) {
    HIPCHK(gen_launch_post(d, &h->parity, h->stream)); h->launches++;
    FIN(7);
    ROWS(0);
    HIPCHK(gen_launch_dual(d, 0, &h->parity, h->stream)); h->launches++;
    return LPBOX_OK;
}
int enqueue_iteration(lpbox_bqp *h, const GenDev &d) {
    HIPCHK(gen_launch_prep(d, 1, &h->parity, h->stream)); h->launches++;
    FIN(1);
    HIPCHK(gen_launch_y(d, &h->parity, h->stream)); h->launches++;
    HIPCHK(gen_launch_rhs_cols(d, &h->parity, h->stream)); h->launches++;
    ROWS(0);
    HIPCHK(gen_launch_resid(d, &h->parity, h->stream)); h->launches++;
    FIN(3);
    CHK(enqueue_pcg(h, d, h->kmax));
    return enqueue_tail(h, d);
}
int read_state(lpbox_bqp *h) {
    HIPCHK(hipMemcpyAsync(&h->hst, h->st.p + h->parity, sizeof(GenState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LPBOX_OK;
}
// The iteration chain is a static launch sequence (device-side control state, fall-through launches), so it is replayed from a
// hipGraph like the large-LP chain: an outer iteration flips the state ping-pong 8 + 3 kmax times, two iterations return to the
// start parity.  Short kernels (n ~ 1e5) are launch-bound without it.
constexpr int GEN_ITERS_PER_GRAPH = 2;
int ensure_graph(lpbox_bqp *h, const GenDev &d, hipGraphExec_t *out) {
    const auto key = std::make_pair(h->kmax, h->parity);
    auto it = h->gexec.find(key);
    if (it != h->gexec.end()) { *out = it->second; return LPBOX_OK; }
    const int par0 = h->parity;
    const long long l0 = h->launches;
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = LPBOX_OK;
    for (int i = 0; i < GEN_ITERS_PER_GRAPH && rc >= 0; i++) rc = enqueue_iteration(h, d);
    const hipError_t e2 = hipStreamEndCapture(h->stream, &g);
    h->graph_launches = h->launches - l0;
    h->launches = l0;
    if (rc < 0 || e2 != hipSuccess || h->parity != par0) {
        h->parity = par0;
        if (g) (void)hipGraphDestroy(g);
        return rc < 0 ? rc : lpbox_fail(LPBOX_E_HIP, "graph capture failed: %s", hipGetErrorString(e2));
    }
    hipGraphExec_t ex = nullptr;
    HIPCHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    h->graphs.push_back(g);
    h->gexec[key] = ex;
    *out = ex;
    return LPBOX_OK;
}
}  // namespace

extern "C" {

lpbox_bqp_t *lpbox_bqp_create(int device) {
    lpbox_bqp *h = new lpbox_bqp();
    h->device = device;
    memset(&h->hst, 0, sizeof(h->hst));
    lpbox_bqp_preset(h, 0);
    if (getenv("LPBOX_BQP_NOGRAPH")) h->use_graph = false;
    return h;
}

void lpbox_bqp_destroy(lpbox_bqp_t *h) {
    if (!h) return;
    if (h->uploaded) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (DevCsr *c : {&h->dA, &h->dCr, &h->dCc, &h->dEr, &h->dEc}) { c->ptr.release(); c->idx.release(); c->val.release(); }
    for (Buf<double> *bp : {&h->tmval, &h->Cc_sv, &h->Ec_sv, &h->x, &h->xt, &h->y1, &h->y2, &h->z1, &h->z2, &h->db, &h->rhs, &h->r, &h->z, &h->tmp, &h->p0,
                            &h->p1, &h->gsrc, &h->pdiag, &h->dinv, &h->Csq, &h->Esq, &h->best, &h->dx0, &h->z3, &h->qC, &h->dd, &h->y3, &h->z4, &h->fy, &h->qE,
                            &h->Ex, &h->df, &h->part, &h->red})
        bp->release();
    h->d_adiag.release(); h->zp.release(); h->st.release();
    for (auto &kv : h->gexec) (void)hipGraphExecDestroy(kv.second);
    for (hipGraph_t g : h->graphs) (void)hipGraphDestroy(g);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int lpbox_bqp_preset(lpbox_bqp_t *h, int type) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    GenParams &p = h->prm;
    switch (type) {
    case 0:      // ADMM_bqp_unconstrained_init SEGcpp:658-672
        p.std_threshold = 1e-6; p.gamma_val = 1.0; p.gamma_factor = 0.99; p.initial_rho = 5; p.learning_fact = 1 + 3.0 / 100; p.history_size = 5;
        p.rho_change_step = 5; p.stop_threshold = 1e-3; p.max_iters = (int)1e4; p.pcg_tol = 1e-3; p.pcg_maxiters = (int)1e3; return LPBOX_OK;
    case 1:      // ADMM_bqp_linear_eq_init :587-601
        p.stop_threshold = 1e-4; p.std_threshold = 1e-6; p.gamma_val = 1.6; p.gamma_factor = 0.95; p.rho_change_step = 5; p.max_iters = (int)5e3;
        p.initial_rho = 1; p.history_size = 3; p.learning_fact = 1 + 5.0 / 100; p.pcg_tol = 1e-4; p.pcg_maxiters = (int)1e3; return LPBOX_OK;
    case 2:      // ADMM_bqp_linear_ineq_init :603-617
    case 3:      // ADMM_bqp_linear_eq_and_uneq_init :620-634
        p.stop_threshold = 1e-4; p.std_threshold = 1e-6; p.gamma_val = 1.6; p.gamma_factor = 0.95; p.rho_change_step = 5; p.max_iters = (int)1e4;
        p.initial_rho = 25; p.history_size = 3; p.learning_fact = 1 + 1.0 / 100; p.pcg_tol = 1e-4; p.pcg_maxiters = (int)1e3; return LPBOX_OK;
    }
    return lpbox_fail(LPBOX_E_BADARG, "preset %d: 0 unconstrained, 1 equality, 2 inequality, 3 both", type);
}

int lpbox_bqp_set_params(lpbox_bqp_t *h, const double *p11) {
    if (!h || !p11) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    GenParams &p = h->prm;
    p.stop_threshold = p11[0]; p.std_threshold = p11[1]; p.gamma_val = p11[2]; p.gamma_factor = p11[3]; p.rho_change_step = (int)p11[4];
    p.max_iters = (int)p11[5]; p.initial_rho = p11[6]; p.history_size = (int)p11[7]; p.learning_fact = p11[8]; p.pcg_tol = p11[9];
    p.pcg_maxiters = (int)p11[10];
    if (p.history_size < 2 || p.history_size > GEN_HIST_MAX) return lpbox_fail(LPBOX_E_BADARG, "history_size must be in [2,%d]", GEN_HIST_MAX);
    if (p.rho_change_step < 1 || p.max_iters < 0 || p.pcg_maxiters < 1) return lpbox_fail(LPBOX_E_BADARG, "bad iteration parameters");
    return LPBOX_OK;
}

int lpbox_bqp_set_problem(lpbox_bqp_t *h, int n, const int *Ap, const int *Ai, const double *Av, const double *b, const double *x0,
                          int m, const int *Cp, const int *Ci, const double *Cv, const double *d,
                          int l, const int *Ep, const int *Ei, const double *Ev, const double *f) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (h->uploaded) return lpbox_fail(LPBOX_E_STATE, "problem already uploaded; create a new handle");
    if (n <= 0 || !b || !x0 || m < 0 || l < 0) return lpbox_fail(LPBOX_E_BADARG, "bad problem arguments");
    CHK(load_csr(h->A, n, n, Ap, Ai, Av, "A"));
    h->adiag.assign(n, -1);
    for (int i = 0; i < n; i++) {
        for (int k = Ap[i]; k < Ap[i + 1]; k++) if (Ai[k] == i) h->adiag[i] = k;
        if (h->adiag[i] < 0)      // `.diagonal() +=` on a compressed sparse matrix needs the entry to exist (SEGcpp:1483): store explicit zeros
            return lpbox_fail(LPBOX_E_BADARG, "A has no stored diagonal entry in row %d", i);
    }
    if (m > 0) { if (!d) return lpbox_fail(LPBOX_E_BADARG, "d missing"); CHK(load_csr(h->C, m, n, Cp, Ci, Cv, "C")); transpose(h->C, h->Ct); h->d.assign(d, d + m); }
    if (l > 0) { if (!f) return lpbox_fail(LPBOX_E_BADARG, "f missing"); CHK(load_csr(h->E, l, n, Ep, Ei, Ev, "E")); transpose(h->E, h->Et); h->f.assign(f, f + l); }
    h->n = n; h->m = m; h->l = l;
    h->b.assign(b, b + n); h->x0.assign(x0, x0 + n);
    h->has_problem = true;
    return LPBOX_OK;
}

int lpbox_bqp_solve(lpbox_bqp_t *h, int *iterations) {                              // ADMM_bqp SEGcpp:1384-1832
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->has_problem) return lpbox_fail(LPBOX_E_STATE, "no problem set");
    CHK(use_device(h));
    const int n = h->n, m = h->m, l = h->l;
    if (!h->uploaded) {
        auto groups = [](int len, int &ept) { ept = 2; while ((len + GEN_T * ept - 1) / (GEN_T * ept) > 4096 && ept < 64) ept *= 2; return len > 0 ? (len + GEN_T * ept - 1) / (GEN_T * ept) : 0; };
        h->G = groups(n, h->EPT); h->Gm = groups(m, h->EPTm); h->Gl = groups(l, h->EPTl);
        HIPCHK(hipStreamCreate(&h->stream));
        HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1));
        HIPCHK(upload(h->dA, h->A)); HIPCHK(h->d_adiag.upload(h->adiag)); HIPCHK(h->tmval.alloc(h->A.val.size()));
        HIPCHK(upload(h->dCr, h->C)); HIPCHK(upload(h->dCc, h->Ct)); HIPCHK(h->Cc_sv.alloc(h->Ct.val.size()));
        HIPCHK(upload(h->dEr, h->E)); HIPCHK(upload(h->dEc, h->Et)); HIPCHK(h->Ec_sv.alloc(h->Et.val.size()));
        for (Buf<double> *bp : {&h->x, &h->xt, &h->y1, &h->y2, &h->z1, &h->z2, &h->rhs, &h->r, &h->z, &h->tmp, &h->p0, &h->p1, &h->gsrc, &h->pdiag, &h->dinv,
                                &h->Csq, &h->Esq, &h->best})
            HIPCHK(bp->alloc(n));
        HIPCHK(h->zp.alloc(n));
        HIPCHK(h->db.upload(h->b)); HIPCHK(h->dx0.upload(h->x0)); HIPCHK(h->dd.upload(h->d)); HIPCHK(h->df.upload(h->f));
        for (Buf<double> *bp : {&h->z3, &h->qC}) HIPCHK(bp->alloc(m));
        for (Buf<double> *bp : {&h->y3, &h->z4, &h->fy, &h->qE, &h->Ex}) HIPCHK(bp->alloc(l));
        HIPCHK(h->part.alloc((size_t)GEN_NPART * h->G)); HIPCHK(h->red.alloc(GEN_NPART)); HIPCHK(h->st.alloc(2));
        HIPCHK(hipMemset(h->part.p, 0, sizeof(double) * (size_t)GEN_NPART * h->G));
        HIPCHK(hipMemset(h->red.p, 0, sizeof(double) * GEN_NPART));
        h->uploaded = true;
    }
    const GenDev d = h->dev();
    h->parity = 0; h->kernel_ms = 0; h->launches = 0;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(gen_launch_init(d, std::pow((double)n, 1.0 / 2), h->dx0.p, h->stream));    // std::pow(n, 1.0 / p), p = 2 (SEGcpp:556)
    HIPCHK(gen_launch_init2(d, h->stream));
    ROWS(0);                                                                        // E x0 for the first y3
    HIPCHK(gen_launch_dual(d, 1, &h->parity, h->stream));
    for (;;) {
        CHK(read_state(h));
        if (h->hst.halt == GEN_HALT_PCG_MORE) {
            HIPCHK(gen_launch_resume(d, 0, &h->parity, h->stream));
            CHK(enqueue_pcg(h, d, 16));
            CHK(enqueue_tail(h, d));
            if (h->adaptive) h->kmax = std::max(h->kmax, h->hst.pcg_k + 8);
            continue;
        }
        if (h->hst.halt != GEN_HALT_NONE) break;
        const int remaining = h->prm.max_iters - h->hst.iter;
        if (remaining <= 0 && !h->hst.have_prev) break;
        if (h->adaptive && h->hst.outer_total > 0) h->kmax = std::max(3, h->hst.pcg_max + 1);      // one spare PCG launch group; the halt-and-resume path covers a miss
        HIPCHK(gen_launch_resume(d, 1, &h->parity, h->stream));
        int batch = std::min(std::max(remaining, 0), 16);
        if (h->use_graph && batch >= GEN_ITERS_PER_GRAPH) {
            hipGraphExec_t ex = nullptr;
            if (ensure_graph(h, d, &ex) < 0) { h->use_graph = false; (void)hipGetLastError(); }
            else
                for (; batch >= GEN_ITERS_PER_GRAPH; batch -= GEN_ITERS_PER_GRAPH) { HIPCHK(hipGraphLaunch(ex, h->stream)); h->launches += h->graph_launches; }
        }
        for (int it = 0; it < batch; it++) CHK(enqueue_iteration(h, d));
        HIPCHK(gen_launch_prep(d, 0, &h->parity, h->stream)); h->launches++;          // finalise the last iteration of the batch
    }
    // best_sol = x_sol of the last improving iteration (:1792) is copied by the NEXT iteration's y kernel; if the loop ended right after an
    // improvement, do it here (x is not touched after the halt)
    if (h->hst.copy_best) HIPCHK(hipMemcpyAsync(h->best.p, h->x.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->kernel_ms = ms;
    h->solved = true;
    if (iterations) *iterations = h->hst.iter;
    return LPBOX_OK;
}

int lpbox_bqp_get_vec(lpbox_bqp_t *h, const char *name, double *out, long cap) {
    if (!h || !h->solved || !out || !name) return lpbox_fail(LPBOX_E_STATE, "not solved");
    CHK(use_device(h));
    const double *src = nullptr; long len = h->n;
    if (!strcmp(name, "x")) src = h->x.p; else if (!strcmp(name, "y1")) src = h->y1.p; else if (!strcmp(name, "y2")) src = h->y2.p;
    else if (!strcmp(name, "z1")) src = h->z1.p; else if (!strcmp(name, "z2")) src = h->z2.p; else if (!strcmp(name, "best_sol")) src = h->best.p;
    else if (!strcmp(name, "z3")) { src = h->z3.p; len = h->m; } else if (!strcmp(name, "z4")) { src = h->z4.p; len = h->l; }
    else if (!strcmp(name, "y3")) { src = h->y3.p; len = h->l; }
    else return lpbox_fail(LPBOX_E_BADARG, "unknown vector '%s'", name);
    if (cap < len) return lpbox_fail(LPBOX_E_BADARG, "buffer too small");
    if (len) HIPCHK(hipMemcpy(out, src, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    return (int)len;
}

int lpbox_bqp_get_scalar(lpbox_bqp_t *h, const char *name, double *out) {
    if (!h || !h->solved || !out || !name) return lpbox_fail(LPBOX_E_STATE, "not solved");
    const GenState &s = h->hst;
    struct { const char *n; double v; } tab[] = {
        {"rho1", s.rho1}, {"rho3", s.rho3}, {"rho4", s.rho4}, {"gamma", s.gamma_val}, {"std_obj", s.std_obj}, {"cvg1", s.cvg1}, {"cvg2", s.cvg2},
        {"cur_obj", s.cur_obj}, {"best_bin_obj", s.best_bin_obj}, {"obj_val", s.obj_val}, {"iters", (double)s.iter}, {"stop", (double)s.stop},
        {"total_pcg", (double)s.pcg_total}, {"outer_total", (double)s.outer_total}, {"last_pcg", (double)s.last_pcg},
        {"kernel_ms", h->kernel_ms}, {"launches", (double)h->launches}, {"threads", (double)GEN_T}, {"chunk", (double)(GEN_T * h->EPT)},
    };
    for (auto &e : tab) if (!strcmp(e.n, name)) { *out = e.v; return LPBOX_OK; }
    return lpbox_fail(LPBOX_E_BADARG, "unknown scalar '%s'", name);
}

}  // extern "C"
