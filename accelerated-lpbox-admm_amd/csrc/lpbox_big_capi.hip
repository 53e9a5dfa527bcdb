// lpbox_big_capi.hip -- host side + C-ABI of the LARGE-instance LP path (lpbox_big_* in include/lpbox_hip.h):
// one LP, variable-sharded over ranks (one process per GPU).  The library launches the kernels; where the algorithm needs a
// sum over all variables it calls the caller-supplied all-reduce (RCCL through torch.distributed in lpbox_hip/big.py) on
// the same HIP stream.  With one rank no collective is issued.
#include "../../include/lpbox_hip.h"
#include "lpbox_big.h"
#include "lpbox_capi_internal.h"

#include <rccl/rccl.h>     // types only: the library is dlopen'ed (no link dependency; a process that has torch loaded shares torch's copy)
#include <dlfcn.h>

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

namespace {
template <typename Tp>
struct Buf {
    Tp *p = nullptr; size_t count = 0;
    hipError_t alloc(size_t c) { release(); count = c; return c ? hipMalloc((void **)&p, c * sizeof(Tp)) : hipSuccess; }
    void release() { if (p) (void)hipFree(p); p = nullptr; count = 0; }
};
// RCCL entry points, resolved at run time
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.lib) return LPBOX_OK;
    const char *names[] = {getenv("LPBOX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *nm : names) if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return lpbox_fail(LPBOX_E_STATE, "cannot load librccl.so: %s", dlerror());
#define RSYM(field, name) do { *(void **)&g_rccl.field = dlsym(lib, name); if (!g_rccl.field) return lpbox_fail(LPBOX_E_STATE, "librccl.so lacks %s", name); } while (0)
    RSYM(GetUniqueId, "ncclGetUniqueId"); RSYM(CommInitRank, "ncclCommInitRank"); RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(AllGather, "ncclAllGather"); RSYM(Send, "ncclSend"); RSYM(Recv, "ncclRecv");
    RSYM(GroupStart, "ncclGroupStart"); RSYM(GroupEnd, "ncclGroupEnd"); RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
    g_rccl.lib = lib;
    return LPBOX_OK;
}
#define NCCLCHK(expr)                                                                                                      \
    do {                                                                                                                   \
        ncclResult_t r_ = (expr);                                                                                          \
        if (r_ != ncclSuccess) return lpbox_fail(LPBOX_E_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));          \
    } while (0)
}  // namespace

struct lpbox_big {
    int rank = 0, world = 1, device = 0;
    long n_glob = 0; int c0 = 0, n_loc = 0, l = 0, nnz = 0;
    std::vector<int> cptr, crow, rptr, rcol;
    std::vector<double> b, f;
    bool has_problem = false, uploaded = false, inited = false, own_stream = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    lpbox_allgather_fn ag = nullptr; void *ag_user = nullptr;   // exchange through the caller (tests: gloo) ...
    ncclComm_t comm = nullptr;                                  // ... or through RCCL, driven from here (no Python in the loop)
    long q_cap = 0;                                             // doubles allocated behind q (l rounded up to a multiple of world)
    // launch-bound inner loop: BIG_ITERS_PER_GRAPH outer iterations captured once per (PCG launch count, start parity) and replayed
    std::map<std::pair<int, int>, hipGraphExec_t> gexec; std::vector<hipGraph_t> graphs;
    long long graph_launches = 0, graph_collectives = 0;       // of ONE replay (measured while capturing)
    bool use_graph = true;
    int G = 0, Gl = 0, EPT = 2, EPTl = 2, P = 1, Glr = 0, kmax = 28, parity = 0;
    bool fold = false; int Gs = 0, Gr[BIG_MAXW] = {0};        // folded reductions (BigDev::fold): partial stride, workgroups of every rank
    Buf<double> gpart;                                          // W > 1, folded: the gathered partials [phase][rank][nv][Gs]
    bool lean = false;                                          // lpbox_big_set_pcg_mode(LPBOX_PCG_COMM_LEAN), before lpbox_big_init
    int Gq = 0, Gqs = 0, Gqr[BIG_MAXW] = {0};                   // q.q partials: of this rank, stride, of every rank (BigDev::Gq)
    Buf<double> lsmall, gsmall;
    bool adaptive = true;
    int kmargin = 1;                                            // spare PCG launch groups beyond the largest count of the previous batch (LPBOX_BIG_KMARGIN)
    double kernel_ms = 0.0; long long launches = 0, collectives = 0;
    Buf<int> d_rptr, d_rcol, d_cptr, d_crow;
    Buf<double> x, y1, y2, z1, z2, db, pd, dinv, rhs, r, z, tmp, p0, p1, gsrc, y3, z4, df, Ex, q, part, red, xt, xhist, xi_out, gath, flag;
    Buf<uint8_t> live, newfix;
    Buf<double2> zp, fz;
    Buf<int> d_live_idx;
    std::vector<int> left_idx, xi_left;   // local indices of the live variables (now / as of the last l2f window)
    long n_live_glob = 0;                 // live variables over all ranks
    int ws_cap = 0, xi_rows = 0;
    bool xi_valid = false;
    bool record = false, xi_plain = false;   // lpbox_big_set_record: the plain loop stages every iterate; the last staged window came from it
    Buf<BigState> st;
    BigState hst;

    BigDev dev() const {
        BigDev d;
        d.n_loc = n_loc; d.l = l; d.G = G; d.Gl = Gl; d.EPT = EPT; d.EPTl = EPTl; d.P = P; d.Glr = Glr; d.n_glob = n_glob;
        d.rptr = d_rptr.p; d.rcol = d_rcol.p; d.cptr = d_cptr.p; d.crow = d_crow.p;
        d.x = x.p; d.y1 = y1.p; d.y2 = y2.p; d.z1 = z1.p; d.z2 = z2.p; d.b = db.p; d.pd = pd.p; d.dinv = dinv.p; d.rhs = rhs.p;
        d.r = r.p; d.z = z.p; d.tmp = tmp.p; d.p0 = p0.p; d.p1 = p1.p; d.gsrc = gsrc.p;
        d.zp = zp.p; d.xt = xt.p; d.live = live.p; d.newfix = newfix.p; d.xhist = xhist.p; d.ws_cap = ws_cap;
        d.y3 = y3.p; d.z4 = z4.p; d.f = df.p; d.fz = fz.p; d.Ex = Ex.p; d.q = q.p; d.part = part.p; d.red = red.p; d.st = st.p;
        d.fold = fold ? 1 : 0; d.W = world; d.Gs = Gs; d.gpart = gpart.p; d.gathered = gpart.p != nullptr;
        for (int r = 0; r < BIG_MAXW; r++) { d.Gr[r] = Gr[r]; d.Gqr[r] = Gqr[r]; }
        d.lean = lean ? 1 : 0; d.Gq = Gq; d.Gqs = Gqs; d.lsmall = lsmall.p; d.gsmall = gsmall.p;
        return d;
    }
};

namespace {

int use_device(lpbox_big *h) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return lpbox_fail(LPBOX_E_NODEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(h->device));
    return LPBOX_OK;
}

// In-place sum of `count` doubles at ptr over all ranks with a FIXED association: element i = ((c0[i] + c1[i]) + c2[i]) + ...
// in rank order, whatever the transport -- so a W-rank run is reproducible and comparable bit for bit with the oracle's model of
// the rank partition (oracle/lpbox_oracle.c lpo_set_ranks).  Transport: RCCL driven from here (vectors: exchange of row blocks
// = reduce-scatter with our own adds, then an all-gather of the reduced blocks; scalars: one all-gather), or the caller's
// all-gather callback (tests: gloo).  One rank without a transport: nothing to do.
int allreduce(lpbox_big *h, double *ptr, long count) {
    if (!h->comm && !h->ag) {
        if (h->world > 1) return lpbox_fail(LPBOX_E_STATE, "world = %d but neither an RCCL communicator nor an all-gather callback was set", h->world);
        return LPBOX_OK;
    }
    const int W = h->world;
    if (h->comm && count > 64) {
        const long lb = (count + W - 1) / W;                        // row block per rank (ptr has room for W * lb doubles)
        if ((long)W * lb > h->q_cap || ptr != h->q.p) return lpbox_fail(LPBOX_E_STATE, "vector exchange outside the padded buffer");
        NCCLCHK(g_rccl.GroupStart());
        for (int pr = 0; pr < W; pr++) {
            NCCLCHK(g_rccl.Send(ptr + (long)pr * lb, (size_t)lb, ncclDouble, pr, h->comm, h->stream));
            NCCLCHK(g_rccl.Recv(h->gath.p + (long)pr * lb, (size_t)lb, ncclDouble, pr, h->comm, h->stream));
        }
        NCCLCHK(g_rccl.GroupEnd());
        HIPCHK(big_launch_rank_sum(h->gath.p, W, lb, lb, ptr + (long)h->rank * lb, h->stream));
        NCCLCHK(g_rccl.AllGather(ptr + (long)h->rank * lb, ptr, (size_t)lb, ncclDouble, h->comm, h->stream));
        h->collectives += 2;
        return LPBOX_OK;
    }
    if (h->comm) NCCLCHK(g_rccl.AllGather(ptr, h->gath.p, (size_t)count, ncclDouble, h->comm, h->stream));
    else {
        const int rc = h->ag(ptr, count, h->gath.p, h->ag_user);
        if (rc != 0) return lpbox_fail(LPBOX_E_HIP, "all-gather callback failed (%d)", rc);
    }
    HIPCHK(big_launch_rank_sum(h->gath.p, W, count, count, ptr, h->stream));
    h->collectives++;
    return LPBOX_OK;
}

// every rank learns whether ANY rank wants to bail out, before a state-changing collective is entered (a rank that returned early
// on its own would leave the others waiting inside the next exchange)
int agree_ok(lpbox_big *h, bool ok_here) {
    if (!h->comm && !h->ag) return ok_here ? LPBOX_OK : -1;
    const double v = ok_here ? 0.0 : 1.0;
    HIPCHK(hipMemcpyAsync(h->flag.p, &v, sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = allreduce(h, h->flag.p, 1);
    if (rc < 0) return rc;
    double tot = 0.0;
    HIPCHK(hipMemcpyAsync(&tot, h->flag.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return tot == 0.0 ? LPBOX_OK : -1;
}

#define CHK(expr) do { int rc_ = (expr); if (rc_ < 0) return rc_; } while (0)
// the totals of a phase's workgroup partials for its consumers.  Folded route: the consumers add them up themselves; only W > 1 ranks
// have something to do here -- ONE all-gather of the partials (nv * Gs doubles per rank), no reduction launch, no rank-sum launch.
// Other route: the reduction launch leaves them in red[], all-reduced over the ranks in rank order.
int gather_partials(lpbox_big *h, const BigDev &d, int phase, int nv) {
    const size_t PS = (size_t)BIG_NPART * h->Gs;
    double *src = h->part.p + (size_t)phase * PS, *dst = h->gpart.p + (size_t)phase * h->world * PS;
    const long count = (long)nv * h->Gs;
    if (h->comm) NCCLCHK(g_rccl.AllGather(src, dst, (size_t)count, ncclDouble, h->comm, h->stream));
    else if (h->ag) {
        const int rc = h->ag(src, count, dst, h->ag_user);
        if (rc != 0) return lpbox_fail(LPBOX_E_HIP, "all-gather callback failed (%d)", rc);
    } else return lpbox_fail(LPBOX_E_STATE, "world = %d but neither an RCCL communicator nor an all-gather callback was set", h->world);
    h->collectives++;
    return LPBOX_OK;
}
#define FIN(nv, phase) do { if (h->fold) { if (h->gpart.p) CHK(gather_partials(h, d, phase, nv)); } \
                            else { HIPCHK(big_launch_fin(d, nv, phase, h->stream)); h->launches++; CHK(allreduce(h, d.red + (phase) * BIG_NPART, nv)); } } while (0)
#define FINX(nv) do { HIPCHK(big_launch_fin(d, nv, BIG_PH_X, h->stream)); h->launches++; CHK(allreduce(h, d.red + BIG_PH_X * BIG_NPART, nv)); } while (0)
#define ROWS(mode) do { HIPCHK(big_launch_rows(d, mode, &h->parity, h->stream)); h->launches++; CHK(allreduce(h, d.q, h->l)); } while (0)

// comm-lean PCG, W > 1: the q exchange of allreduce() with the scalars riding along -- the row blocks go to their owners (one grouped
// send/recv), the owner adds its block in rank order and squares it (big_k_rank_sum_qq: q.q partials behind the p.p partials the row
// kernel left in lsmall), then ONE grouped all-gather brings back the reduced blocks and everybody's partials.
int exchange_q_lean(lpbox_big *h, const BigDev &d) {
    const int W = h->world;
    const long lb = ((long)h->l + W - 1) / W;
    const long lo = std::min<long>((long)h->rank * lb, h->l), mine = std::min<long>(lb, h->l - lo);
    const size_t small = (size_t)h->Gs + h->Gqs;
    if (h->comm) {
        NCCLCHK(g_rccl.GroupStart());
        for (int pr = 0; pr < W; pr++) {
            NCCLCHK(g_rccl.Send(h->q.p + (long)pr * lb, (size_t)lb, ncclDouble, pr, h->comm, h->stream));
            NCCLCHK(g_rccl.Recv(h->gath.p + (long)pr * lb, (size_t)lb, ncclDouble, pr, h->comm, h->stream));
        }
        NCCLCHK(g_rccl.GroupEnd());
        HIPCHK(big_launch_rank_sum_qq(h->gath.p, W, mine, lb, h->q.p + (long)h->rank * lb, h->lsmall.p + h->Gs, h->stream));
        NCCLCHK(g_rccl.GroupStart());
        NCCLCHK(g_rccl.AllGather(h->q.p + (long)h->rank * lb, h->q.p, (size_t)lb, ncclDouble, h->comm, h->stream));
        NCCLCHK(g_rccl.AllGather(h->lsmall.p, h->gsmall.p, small, ncclDouble, h->comm, h->stream));
        NCCLCHK(g_rccl.GroupEnd());
        h->collectives += 2;
        return LPBOX_OK;
    }
    if (!h->ag) return lpbox_fail(LPBOX_E_STATE, "world = %d but neither an RCCL communicator nor an all-gather callback was set", W);
    // callback transport (tests): whole contributions are gathered, every rank adds all rows in rank order; only its own block's squares count
    int rc = h->ag(h->q.p, h->l, h->gath.p, h->ag_user);
    if (rc != 0) return lpbox_fail(LPBOX_E_HIP, "all-gather callback failed (%d)", rc);
    HIPCHK(big_launch_rank_sum(h->gath.p, W, h->l, h->l, h->q.p, h->stream));
    HIPCHK(big_launch_rank_sum_qq(h->gath.p + lo, W, mine, h->l, h->q.p + lo, h->lsmall.p + h->Gs, h->stream));
    rc = h->ag(h->lsmall.p, (long)small, h->gsmall.p, h->ag_user);
    if (rc != 0) return lpbox_fail(LPBOX_E_HIP, "all-gather callback failed (%d)", rc);
    h->collectives += 2;
    return LPBOX_OK;
}

int enqueue_pcg(lpbox_big *h, const BigDev &d, int pairs) {
    if (h->lean) {
        for (int k = 0; k < pairs; k++) {
            HIPCHK(big_launch_rows(d, 1, &h->parity, h->stream)); h->launches++;
            if (h->world > 1 || h->comm) CHK(exchange_q_lean(h, d));
            HIPCHK(big_launch_pcg_lean(d, &h->parity, h->stream)); h->launches++;
            FIN(2, BIG_PH_D);
        }
        return LPBOX_OK;
    }
    for (int k = 0; k < pairs; k++) {
        ROWS(1);
        HIPCHK(big_launch_pcg_cols(d, &h->parity, h->stream)); h->launches++;
        FIN(1, BIG_PH_C);
        HIPCHK(big_launch_pcg_upd(d, &h->parity, h->stream)); h->launches++;
        FIN(2, BIG_PH_D);
    }
    return LPBOX_OK;
}

int enqueue_tail(lpbox_big *h, const BigDev &d) {
    HIPCHK(big_launch_post(d, &h->parity, h->stream)); h->launches++;
    FIN(5, BIG_PH_E);
    ROWS(0);
    HIPCHK(big_launch_z4(d, 0, &h->parity, h->stream)); h->launches++;
    return LPBOX_OK;
}

int enqueue_iteration(lpbox_big *h, const BigDev &d) {
    HIPCHK(big_launch_prep(d, 1, &h->parity, h->stream)); h->launches++;
    FIN(1, BIG_PH_A);
    HIPCHK(big_launch_y(d, &h->parity, h->stream)); h->launches++;
    HIPCHK(big_launch_rhs_cols(d, &h->parity, h->stream)); h->launches++;
    ROWS(0);
    HIPCHK(big_launch_resid(d, &h->parity, h->stream)); h->launches++;
    FIN(3, BIG_PH_B);
    CHK(enqueue_pcg(h, d, h->kmax));
    return enqueue_tail(h, d);
}

int read_state(lpbox_big *h) {
    HIPCHK(hipMemcpyAsync(&h->hst, h->st.p + h->parity, sizeof(BigState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LPBOX_OK;
}

// An outer iteration is 8 + 3*kmax launches that flip the state ping-pong, so two iterations return to the parity they started
// from: one hipGraph = BIG_ITERS_PER_GRAPH iterations per (kmax, start parity).  The caller's all-gather callback cannot be
// captured (it runs host code at enqueue time); RCCL operations and the single-rank chain can.
constexpr int BIG_ITERS_PER_GRAPH = 2;
int ensure_graph(lpbox_big *h, const BigDev &d, hipGraphExec_t *out) {
    const auto key = std::make_pair(h->kmax, h->parity);
    auto it = h->gexec.find(key);
    if (it != h->gexec.end()) { *out = it->second; return LPBOX_OK; }
    const int par0 = h->parity;
    const long long l0 = h->launches, c0 = h->collectives;
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = LPBOX_OK;
    for (int i = 0; i < BIG_ITERS_PER_GRAPH && rc >= 0; i++) rc = enqueue_iteration(h, d);
    const hipError_t e2 = hipStreamEndCapture(h->stream, &g);
    h->graph_launches = h->launches - l0; h->graph_collectives = h->collectives - c0;
    h->launches = l0; h->collectives = c0;
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

lpbox_big_t *lpbox_big_create(int rank, int world, int device) {
    if (world < 1 || rank < 0 || rank >= world) { lpbox_fail(LPBOX_E_BADARG, "bad rank/world %d/%d", rank, world); return nullptr; }
    lpbox_big *h = new lpbox_big();
    h->rank = rank; h->world = world; h->device = device;
    memset(&h->hst, 0, sizeof(h->hst));
    if (getenv("LPBOX_BIG_NOGRAPH")) h->use_graph = false;                        // eager launches (profilers that dislike graphs)
    if (const char *e = getenv("LPBOX_BIG_KMAX")) { int v = atoi(e); if (v >= 1 && v <= 1000) { h->kmax = v; h->adaptive = false; } }
    return h;
}

void lpbox_big_destroy(lpbox_big_t *h) {
    if (!h) return;
    if (h->uploaded) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->d_rptr.release(); h->d_rcol.release(); h->d_cptr.release(); h->d_crow.release();
    for (Buf<double> *bp : {&h->x, &h->y1, &h->y2, &h->z1, &h->z2, &h->db, &h->pd, &h->dinv, &h->rhs, &h->r, &h->z, &h->tmp, &h->p0, &h->p1,
                            &h->gsrc, &h->y3, &h->z4, &h->df, &h->Ex, &h->q, &h->part, &h->red, &h->xt, &h->xhist, &h->xi_out, &h->gath, &h->flag, &h->gpart, &h->lsmall, &h->gsmall})
        bp->release();
    for (auto &kv : h->gexec) (void)hipGraphExecDestroy(kv.second);
    for (hipGraph_t g : h->graphs) (void)hipGraphDestroy(g);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->st.release(); h->live.release(); h->newfix.release(); h->d_live_idx.release(); h->zp.release(); h->fz.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int lpbox_big_set_stream(lpbox_big_t *h, void *hip_stream) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (h->own_stream && h->stream) { (void)hipStreamDestroy(h->stream); h->own_stream = false; }
    h->stream = (hipStream_t)hip_stream;
    return LPBOX_OK;
}

// The plain loop's per-iteration dump (print_fix_info 2, LPcpp:777-780, :903-909) behind the size hand-over: the next lpbox_big_iterate
// calls keep x after every iteration in the staging buffer the l2f window uses ([iterations][n_loc] on the device); read it back with
// lpbox_big_get_x_iters(ws = iterations of the call).  Refused per call when the buffer would exceed 4 GiB.
int lpbox_big_set_record(lpbox_big_t *h, int on) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    h->record = on != 0;
    return LPBOX_OK;
}

// Opt-in, NOT the reference's arithmetic: the PCG's step length from p.Mp = dI (p.p) + r4Et (q.q) with q = E p, which
// lets the p.p partials ride with the q exchange and fuses the column product with the vector updates -- 3 instead of 4 RCCL operations and
// 2 instead of 3 kernels per PCG iteration.  Before lpbox_big_init; needs the folded reductions (<= 1024 workgroups, <= 16 ranks).
int lpbox_big_set_pcg_mode(lpbox_big_t *h, int mode) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (h->inited) return lpbox_fail(LPBOX_E_STATE, "set the PCG mode before lpbox_big_init");
    if (mode != LPBOX_PCG_REFERENCE && mode != LPBOX_PCG_COMM_LEAN) return lpbox_fail(LPBOX_E_BADARG, "unknown PCG mode %d", mode);
    h->lean = mode == LPBOX_PCG_COMM_LEAN;
    return LPBOX_OK;
}

int lpbox_big_set_allgather(lpbox_big_t *h, lpbox_allgather_fn fn, void *user) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    // the exchange buffers are sized by lpbox_big_init from the transport that is set then (as for lpbox_big_rccl_init)
    if (h->inited) return lpbox_fail(LPBOX_E_STATE, "set the all-gather callback before lpbox_big_init");
    h->ag = fn; h->ag_user = user;
    return LPBOX_OK;
}

int lpbox_big_rccl_unique_id(void *out128) {
    if (!out128) return lpbox_fail(LPBOX_E_BADARG, "null output");
    int rc = rccl_load();
    if (rc < 0) return rc;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return (int)sizeof(id);
}

int lpbox_big_rccl_init(lpbox_big_t *h, const void *unique_id128) {
    if (!h || !unique_id128) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle / id");
    if (h->comm) return lpbox_fail(LPBOX_E_STATE, "communicator already created");
    if (h->inited) return lpbox_fail(LPBOX_E_STATE, "the communicator must be created before solve_init");
    int rc = rccl_load();
    if (rc < 0) return rc;
    rc = use_device(h);
    if (rc < 0) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    NCCLCHK(g_rccl.CommInitRank(&h->comm, h->world, id, h->rank));
    return LPBOX_OK;
}

int lpbox_big_set_problem(lpbox_big_t *h, long n_glob, int c0, int n_loc, int l, const int *colptr, const int *rowidx,
                          const double *b, const double *f) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (h->uploaded) return lpbox_fail(LPBOX_E_STATE, "problem already uploaded");
    if (n_glob <= 0 || n_loc <= 0 || l <= 0 || c0 < 0 || (long)c0 + n_loc > n_glob || !colptr || !b || colptr[0] != 0)
        return lpbox_fail(LPBOX_E_BADARG, "bad problem arguments");
    const int nnz = colptr[n_loc];
    if (nnz < 0 || (nnz > 0 && !rowidx)) return lpbox_fail(LPBOX_E_BADARG, "row indices missing");
    for (int j = 0; j < n_loc; j++) {
        if (colptr[j] < 0 || colptr[j + 1] < colptr[j] || colptr[j + 1] > nnz) return lpbox_fail(LPBOX_E_BADARG, "colptr not monotone inside [0, nnz]");
        for (int k = colptr[j]; k < colptr[j + 1]; k++) {
            if (rowidx[k] < 0 || rowidx[k] >= l) return lpbox_fail(LPBOX_E_BADARG, "row index out of range");
            if (k > colptr[j] && rowidx[k] <= rowidx[k - 1]) return lpbox_fail(LPBOX_E_BADARG, "row indices must ascend inside a column");
        }
    }
    h->n_glob = n_glob; h->c0 = c0; h->n_loc = n_loc; h->l = l; h->nnz = nnz;
    h->cptr.assign(colptr, colptr + n_loc + 1); h->crow.assign(rowidx, rowidx + nnz);
    // Row-side storage, slice-major (lpbox_big_kernels.hip row_sum_sliced): the columns are cut into P slices of SW columns so that
    // one slice of the gathered (z, p) table (16 B per variable) stays resident in an XCD's L2; run (slice ph, row i) starts at
    // rptr[ph * l + i] and the runs follow each other.  P = 1 (a shard that fits anyway, or LPBOX_BIG_SLICE_KB=0) is plain CSR.
    long slice_kb = 2048;
    if (const char *e = getenv("LPBOX_BIG_SLICE_KB")) slice_kb = atol(e);
    long SW = slice_kb > 0 ? std::max<long>(1024, slice_kb * 1024 / 16) : (long)n_loc;
    int P = (int)std::min<long>(((long)n_loc + SW - 1) / SW, 64);
    if (P <= 1) { P = 1; SW = n_loc; } else SW = ((long)n_loc + P - 1) / P;
    if ((size_t)P * (size_t)l + 1 > (size_t)INT32_MAX) { P = 1; SW = n_loc; }
    h->P = P;
    h->rptr.assign((size_t)P * l + 1, 0);
    for (int j = 0; j < n_loc; j++) {
        const size_t base = (size_t)(j / SW) * l + 1;
        for (int k = colptr[j]; k < colptr[j + 1]; k++) h->rptr[base + rowidx[k]]++;
    }
    for (size_t i = 0; i < (size_t)P * l; i++) h->rptr[i + 1] += h->rptr[i];
    h->rcol.assign(nnz, 0);
    std::vector<int> cur(h->rptr.begin(), h->rptr.end() - 1);
    for (int j = 0; j < n_loc; j++) {
        const size_t base = (size_t)(j / SW) * l;
        for (int k = colptr[j]; k < colptr[j + 1]; k++) h->rcol[cur[base + rowidx[k]]++] = j;
    }
    h->b.assign(b, b + n_loc);
    if (f) h->f.assign(f, f + l); else h->f.assign(l, 1.0);
    h->has_problem = true;
    return LPBOX_OK;
}

int lpbox_big_init(lpbox_big_t *h) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->has_problem) return lpbox_fail(LPBOX_E_STATE, "no problem set");
    CHK(use_device(h));
    if (!h->uploaded) {
        const int n = h->n_loc, l = h->l;
        if (!h->stream) { HIPCHK(hipStreamCreate(&h->stream)); h->own_stream = true; }
        HIPCHK(h->flag.alloc(64));
        h->q_cap = (long)h->world * (((long)l + h->world - 1) / h->world);           // whole row blocks for the exchange
        if (h->comm) HIPCHK(h->gath.alloc((size_t)std::max<long>(h->q_cap, 64L * h->world)));          // W row blocks, or W scalar groups
        else if (h->ag) HIPCHK(h->gath.alloc((size_t)h->world * (size_t)std::max(l, 64)));              // W whole contributions
        // every rank learns the shard sizes of all ranks: the slots per thread (hence the partial stride) must be the same everywhere
        std::vector<double> nloc_all((size_t)std::max(h->world, 1), 0.0);
        nloc_all[h->rank] = (double)n;
        if (h->world > 1 && h->world <= 64) {
            HIPCHK(hipMemcpyAsync(h->flag.p, nloc_all.data(), sizeof(double) * (size_t)h->world, hipMemcpyHostToDevice, h->stream));
            CHK(allreduce(h, h->flag.p, h->world));
            HIPCHK(hipMemcpyAsync(nloc_all.data(), h->flag.p, sizeof(double) * (size_t)h->world, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        long nmax = n;
        for (double v : nloc_all) nmax = std::max(nmax, (long)v);
        // slots per thread: as few as keep the workgroup partials within reach of the folded reductions (G <= BIG_FOLD_U * BIG_T: the
        // consumers add them up themselves, no reduction launch; 10^6 variables on one rank: 4 slots, 977 workgroups), beyond that the
        // reduction launch with at most 4096 partials
        const int fold_cap = BIG_FOLD_U * BIG_T;
        auto groups = [](long nn, int ept) { return (int)((nn + (long)BIG_T * ept - 1) / ((long)BIG_T * ept)); };
        h->EPT = 2; while (groups(nmax, h->EPT) > fold_cap && h->EPT < 8) h->EPT *= 2;
        if (groups(nmax, h->EPT) > fold_cap) { h->EPT = 2; while (groups(nmax, h->EPT) > 2 * BIG_T * 8 && h->EPT < 64) h->EPT *= 2; }
        if (const char *e = getenv("LPBOX_BIG_EPT")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) h->EPT = std::max(v, h->EPT); }   // tuning
        h->G = groups(n, h->EPT);
        h->Gs = std::max(groups(nmax, h->EPT), 1);
        h->fold = h->Gs <= fold_cap && h->world <= BIG_MAXW && getenv("LPBOX_BIG_NOFOLD") == nullptr;   // LPBOX_BIG_NOFOLD: keep the reduction launches (A/B)
        for (int r = 0; r < BIG_MAXW; r++) h->Gr[r] = r < h->world ? groups((long)nloc_all[r], h->EPT) : 0;
        h->EPTl = 2; h->Gl = (l + BIG_T * h->EPTl - 1) / (BIG_T * h->EPTl);
        h->Glr = (l + BIG_T - 1) / BIG_T;
        if (const char *e = getenv("LPBOX_BIG_KMARGIN")) h->kmargin = std::max(0, atoi(e));
        HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1));
        HIPCHK(h->d_rptr.alloc((size_t)h->P * l + 1)); HIPCHK(h->d_rcol.alloc(h->nnz)); HIPCHK(h->d_cptr.alloc((size_t)n + 1)); HIPCHK(h->d_crow.alloc(h->nnz));
        for (Buf<double> *bp : {&h->x, &h->y1, &h->y2, &h->z1, &h->z2, &h->db, &h->pd, &h->dinv, &h->rhs, &h->r, &h->z, &h->tmp, &h->p0, &h->p1, &h->gsrc, &h->xt})
            HIPCHK(bp->alloc(n));
        HIPCHK(h->live.alloc(n)); HIPCHK(h->newfix.alloc(n)); HIPCHK(h->d_live_idx.alloc(n)); HIPCHK(h->zp.alloc(n));
        HIPCHK(hipMemset(h->newfix.p, 0, (size_t)n));
        for (Buf<double> *bp : {&h->y3, &h->z4, &h->df, &h->Ex}) HIPCHK(bp->alloc(l));
        HIPCHK(h->fz.alloc(l));
        HIPCHK(h->q.alloc((size_t)h->q_cap)); HIPCHK(hipMemset(h->q.p, 0, sizeof(double) * (size_t)h->q_cap));
        HIPCHK(h->part.alloc((size_t)BIG_PH_COUNT * BIG_NPART * h->Gs)); HIPCHK(h->red.alloc(BIG_PH_COUNT * BIG_NPART)); HIPCHK(h->st.alloc(2));
        if (h->lean) {
            if (!h->fold) return lpbox_fail(LPBOX_E_UNSUPPORTED, "the comm-lean PCG needs the folded reductions (at most 1024 workgroups per rank, 16 ranks)");
            const int W = h->world;
            const long lb = ((long)l + W - 1) / W;
            const bool xch = W > 1 || h->comm;              // the q exchange runs (also against a one-rank communicator: tests)
            if (xch) {
                for (int r = 0; r < W; r++) { const long lo = std::min<long>((long)r * lb, l), cnt = std::min<long>(lb, l - lo); h->Gqr[r] = (int)((cnt + BIG_T - 1) / BIG_T); }
                h->Gq = h->Gqr[h->rank]; h->Gqs = (int)((lb + BIG_T - 1) / BIG_T);
            } else { h->Gq = h->P > 1 ? h->Glr : h->Gl; h->Gqs = h->Gq; h->Gqr[0] = h->Gq; }
            if (h->Gq > 8 * BIG_T) return lpbox_fail(LPBOX_E_UNSUPPORTED, "the comm-lean PCG sums at most %d row-workgroup partials per rank", 8 * BIG_T);
            HIPCHK(h->lsmall.alloc((size_t)h->Gs + h->Gqs)); HIPCHK(hipMemset(h->lsmall.p, 0, sizeof(double) * ((size_t)h->Gs + h->Gqs)));
            if (xch) { HIPCHK(h->gsmall.alloc((size_t)W * ((size_t)h->Gs + h->Gqs))); HIPCHK(hipMemset(h->gsmall.p, 0, sizeof(double) * (size_t)W * ((size_t)h->Gs + h->Gqs))); }
        }
        if (h->fold && (h->world > 1 || h->comm)) {         // (a one-rank RCCL communicator runs the exchange against itself: tests)
            HIPCHK(h->gpart.alloc((size_t)BIG_PH_COUNT * h->world * BIG_NPART * h->Gs));
            HIPCHK(hipMemset(h->gpart.p, 0, sizeof(double) * (size_t)BIG_PH_COUNT * h->world * BIG_NPART * h->Gs));
        }
        HIPCHK(hipMemcpy(h->d_rptr.p, h->rptr.data(), sizeof(int) * ((size_t)h->P * l + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_rcol.p, h->rcol.data(), sizeof(int) * (size_t)h->nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_cptr.p, h->cptr.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_crow.p, h->crow.data(), sizeof(int) * (size_t)h->nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->db.p, h->b.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(h->part.p, 0, sizeof(double) * (size_t)BIG_PH_COUNT * BIG_NPART * h->Gs));
        HIPCHK(hipMemset(h->red.p, 0, sizeof(double) * BIG_PH_COUNT * BIG_NPART));
        h->uploaded = true;
    }
    HIPCHK(hipMemcpyAsync(h->df.p, h->f.data(), sizeof(double) * (size_t)h->l, hipMemcpyHostToDevice, h->stream));
    h->left_idx.resize(h->n_loc);
    for (int j = 0; j < h->n_loc; j++) h->left_idx[j] = j;
    h->n_live_glob = h->n_glob; h->xi_valid = false;
    const BigDev d = h->dev();
    h->parity = 0;
    HIPCHK(big_launch_init(d, std::pow((double)h->n_glob, 1.0 / 2), h->stream));   // pow(n, 1/p), p = 2 (LPcpp:427,503), n = ALL variables
    CHK(allreduce(h, d.red + BIG_PH_X * BIG_NPART, 1));
    HIPCHK(big_launch_init2(d, h->stream));
    ROWS(0);                                                                        // E * x0 for the first y3 (:720)
    HIPCHK(big_launch_z4(d, 1, &h->parity, h->stream));
    CHK(read_state(h));
    h->inited = true;
    return 1;
}

static int run_window(lpbox_big *h, const BigDev &d, int iter_end) {
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (;;) {
        CHK(read_state(h));
        if (h->hst.halt == BIG_HALT_PCG_MORE) {
            HIPCHK(big_launch_resume(d, 0, &h->parity, h->stream));
            CHK(enqueue_pcg(h, d, 16));
            CHK(enqueue_tail(h, d));
            if (h->adaptive) h->kmax = std::max(h->kmax, h->hst.pcg_k + 8);
            continue;
        }
        if (h->hst.halt != BIG_HALT_NONE) break;
        const int remaining = iter_end - h->hst.iter;
        if (remaining <= 0 && !h->hst.have_prev) break;
        if (h->adaptive && h->hst.outer_total > 0) h->kmax = std::max(4, h->hst.pcg_max + h->kmargin);
        HIPCHK(big_launch_resume(d, 1, &h->parity, h->stream));
        int batch = std::min(std::max(remaining, 0), 16);
        if (h->use_graph && !h->ag && batch >= BIG_ITERS_PER_GRAPH) {
            hipGraphExec_t ex = nullptr;
            if (ensure_graph(h, d, &ex) < 0) {          // a transport that cannot be captured: go on with eager launches for good
                h->use_graph = false;
                (void)hipGetLastError();
            } else {
                for (; batch >= BIG_ITERS_PER_GRAPH; batch -= BIG_ITERS_PER_GRAPH) {
                    HIPCHK(hipGraphLaunch(ex, h->stream));
                    h->launches += h->graph_launches; h->collectives += h->graph_collectives;
                }
            }
        }
        for (int it = 0; it < batch; it++) CHK(enqueue_iteration(h, d));
        HIPCHK(big_launch_prep(d, 0, &h->parity, h->stream)); h->launches++;          // finalise the last iteration of the batch
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->kernel_ms += ms;
    return LPBOX_OK;
}

int lpbox_big_iterate(lpbox_big_t *h, int iter_start, int iter_end, int *ret) {     // ADMM_lp_iters LPcpp:766-1095
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->inited) return lpbox_fail(LPBOX_E_STATE, "solve_init has not been called");
    CHK(use_device(h));
    const int ws = iter_end - iter_start;
    const bool rec = h->record && ws > 0;
    if (rec) {
        if ((double)ws * h->n_loc * sizeof(double) > 4294967296.0)
            return lpbox_fail(LPBOX_E_UNSUPPORTED, "recording %d iterations of %d variables needs more than 4 GiB", ws, h->n_loc);
        if (h->ws_cap < ws || !h->xhist.p) {
            HIPCHK(hipStreamSynchronize(h->stream));
            HIPCHK(h->xhist.alloc((size_t)ws * h->n_loc));
            for (auto &kv : h->gexec) (void)hipGraphExecDestroy(kv.second);         // the captured launches carry the old buffer
            h->gexec.clear();
            h->ws_cap = ws;
        }
        HIPCHK(hipMemsetAsync(h->xhist.p, 0, sizeof(double) * (size_t)h->ws_cap * h->n_loc, h->stream));
        h->xi_left = h->left_idx; h->xi_rows = (int)h->left_idx.size();
        if (h->xi_rows) HIPCHK(hipMemcpyAsync(h->d_live_idx.p, h->xi_left.data(), sizeof(int) * (size_t)h->xi_rows, hipMemcpyHostToDevice, h->stream));
    }
    const BigDev d = h->dev();
    HIPCHK(big_launch_set_window(d, iter_start, iter_end, rec ? 2 : 0, &h->parity, h->stream));
    CHK(run_window(h, d, iter_end));
    if (rec) { h->xi_valid = true; h->xi_plain = true; }
    if (ret) *ret = h->hst.ret;
    return LPBOX_OK;
}

// ADMM_lp_iters_l2f (LPcpp:1098-1574).  vec_local: this rank's slice of the fix vector, one entry per LOCAL live variable in
// ascending order (1 / 0 = fix, anything else = leave); num_global: fixes over all ranks (the caller sums the local counts).
int lpbox_big_iterate_l2f(lpbox_big_t *h, int iter_start, int iter_end, const double *vec_local, long num_global, int *ret) {
    if (!h) return lpbox_fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->inited) return lpbox_fail(LPBOX_E_STATE, "solve_init has not been called");
    const int ws = iter_end - iter_start;
    if (ws > LP_XITERS_COLS) return lpbox_fail(LPBOX_E_BADARG, "window of %d iterations exceeds the %d columns of x_iters (LPcpp:1113)", ws, LP_XITERS_COLS);
    if (num_global < 0 || num_global > h->n_live_glob) return lpbox_fail(LPBOX_E_BADARG, "fix count %ld outside [0,%ld]", num_global, h->n_live_glob);
    CHK(use_device(h));
    const int n_live_loc = (int)h->left_idx.size();
    std::vector<uint8_t> nf;
    std::vector<int> keep;
    bool ok = true;
    if (num_global != 0) {
        if (!vec_local && n_live_loc) { lpbox_fail(LPBOX_E_BADARG, "fix vector missing"); ok = false; }
        else {
            nf.assign(h->n_loc, 0);
            keep.reserve(n_live_loc);
            long cnt = 0;
            for (int q = 0; q < n_live_loc; q++) {
                const int j = h->left_idx[q];
                if (vec_local[q] == 1) { nf[j] = 2; cnt++; } else if (vec_local[q] == 0) { nf[j] = 1; cnt++; } else keep.push_back(j);
            }
            if (h->world == 1 && cnt != num_global) { lpbox_fail(LPBOX_E_BADARG, "vec fixes %ld variables but num = %ld", cnt, num_global); ok = false; }
            else if (cnt > num_global) { lpbox_fail(LPBOX_E_BADARG, "this rank fixes %ld variables, more than the global count %ld", cnt, num_global); ok = false; }
        }
    }
    // all ranks fail together: nobody enters the state-changing collectives below unless everybody passed validation
    if (agree_ok(h, ok) != LPBOX_OK) return ok ? lpbox_fail(LPBOX_E_BADARG, "another rank rejected its arguments of this call") : LPBOX_E_BADARG;
    if (num_global != 0) h->left_idx.swap(keep);
    if (ws > 0 && (h->ws_cap < ws || !h->xhist.p)) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(h->xhist.alloc((size_t)ws * h->n_loc));
        for (auto &kv : h->gexec) (void)hipGraphExecDestroy(kv.second);         // the captured launches carry the old window buffer
        h->gexec.clear();
        h->ws_cap = ws;
    }
    const BigDev d = h->dev();
    HIPCHK(big_launch_set_window(d, iter_start, iter_end, 1, &h->parity, h->stream));
    if (num_global != 0) {
        const long n_live_new = h->n_live_glob - num_global;
        HIPCHK(hipMemcpyAsync(h->newfix.p, nf.data(), nf.size(), hipMemcpyHostToDevice, h->stream));
        HIPCHK(big_launch_fix1(d, h->stream)); h->launches++;
        FINX(1);                                                                    // fix_obj = b2.x2
        ROWS(0);                                                                    // q = E2 * x2
        HIPCHK(big_launch_fix2(d, &h->parity, h->stream)); h->launches++;
        FINX(1);                                                                    // |x_live|^2
        HIPCHK(big_launch_fix3(d, n_live_new, std::pow((double)n_live_new, 1.0 / 2), &h->parity, h->stream)); h->launches++;
        if (n_live_new != 0) {
            ROWS(0);                                                                // E * x (live columns) for the first y3
            HIPCHK(big_launch_z4(d, 1, &h->parity, h->stream)); h->launches++;
        }
        HIPCHK(hipMemsetAsync(h->newfix.p, 0, (size_t)h->n_loc, h->stream));
        h->n_live_glob = n_live_new;
    }
    h->xi_left = h->left_idx; h->xi_rows = (int)h->left_idx.size();
    if (h->ws_cap > 0) HIPCHK(hipMemsetAsync(h->xhist.p, 0, sizeof(double) * (size_t)h->ws_cap * h->n_loc, h->stream));   // x_iters = Zero (:1113): ALL staged columns, also those of an earlier, longer window
    if (h->xi_rows) HIPCHK(hipMemcpyAsync(h->d_live_idx.p, h->xi_left.data(), sizeof(int) * (size_t)h->xi_rows, hipMemcpyHostToDevice, h->stream));
    CHK(run_window(h, d, iter_end));        // synchronises: nf / xi_left stay alive until here
    h->xi_valid = true; h->xi_plain = false;
    if (ret) *ret = h->hst.ret;
    return LPBOX_OK;
}

int lpbox_big_get_n(lpbox_big_t *h) {                                               // live variables of this rank
    if (!h || !h->inited) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    return (int)h->left_idx.size();
}

// (rows x ws) row-major x_iters of this rank's live variables, left on the device; rows = live variables when the window started
int lpbox_big_get_x_iters_device(lpbox_big_t *h, int ws, void **dev_ptr, int *rows) {
    if (!h || !h->inited) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    if (!h->xi_valid) return lpbox_fail(LPBOX_E_STATE, "solve_iter_l2f has not been called");
    const int ws_max = h->xi_plain ? h->ws_cap : LP_XITERS_COLS;                 // a recorded plain call may be longer than an l2f window
    if (ws <= 0 || ws > ws_max) return lpbox_fail(LPBOX_E_BADARG, "ws = %d outside (0,%d]", ws, ws_max);
    CHK(use_device(h));
    const size_t need = (size_t)std::max(h->xi_rows, 1) * ws;
    if (h->xi_out.count < need) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(h->xi_out.alloc(need)); }
    HIPCHK(big_launch_pack_xiters(h->dev(), h->d_live_idx.p, h->xi_rows, ws, h->xi_out.p, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (dev_ptr) *dev_ptr = h->xi_out.p;
    if (rows) *rows = h->xi_rows;
    return LPBOX_OK;
}

int lpbox_big_get_x_iters(lpbox_big_t *h, int ws, double *out) {
    void *p = nullptr; int rows = 0;
    if (!out) { if (!h || !h->xi_valid) return lpbox_fail(LPBOX_E_STATE, "solve_iter_l2f has not been called"); return h->xi_rows; }
    CHK(lpbox_big_get_x_iters_device(h, ws, &p, &rows));
    if (rows) HIPCHK(hipMemcpy(out, p, sizeof(double) * (size_t)rows * ws, hipMemcpyDeviceToHost));
    return rows;
}

// binary solution of this rank's variables: fixed ones at their value, live ones rounded (get_x_sol, LPcpp:1648-1666)
int lpbox_big_get_x_sol(lpbox_big_t *h, double *out_local) {
    if (!h || !h->inited || !out_local) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    CHK(use_device(h));
    std::vector<uint8_t> lv(h->n_loc);
    HIPCHK(hipMemcpy(out_local, h->x.p, sizeof(double) * (size_t)h->n_loc, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lv.data(), h->live.p, (size_t)h->n_loc, hipMemcpyDeviceToHost));
    for (int j = 0; j < h->n_loc; j++) if (lv[j]) out_local[j] = out_local[j] >= 0.5 ? 1.0 : 0.0;
    return h->n_loc;
}

int lpbox_big_cal_obj(lpbox_big_t *h, double *out) {                                // cal_Obj, LPcpp:1630-1642
    if (!h || !h->inited || !out) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    *out = h->n_live_glob != 0 ? h->hst.sum_fix_obj + h->hst.cur_obj : h->hst.sum_fix_obj;
    return LPBOX_OK;
}

int lpbox_big_get_x(lpbox_big_t *h, double *out_local) {
    if (!h || !h->inited || !out_local) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    CHK(use_device(h));
    HIPCHK(hipMemcpy(out_local, h->x.p, sizeof(double) * (size_t)h->n_loc, hipMemcpyDeviceToHost));
    return h->n_loc;
}

int lpbox_big_get_vec(lpbox_big_t *h, const char *name, double *out, long cap) {
    if (!h || !h->inited || !out || !name) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    CHK(use_device(h));
    const double *src = nullptr; long len = h->n_loc;
    if (!strcmp(name, "x")) src = h->x.p; else if (!strcmp(name, "z1")) src = h->z1.p; else if (!strcmp(name, "z2")) src = h->z2.p;
    else if (!strcmp(name, "pd")) src = h->pd.p; else if (!strcmp(name, "b")) src = h->db.p;
    else if (!strcmp(name, "f")) { src = h->df.p; len = h->l; }
    else if (!strcmp(name, "z4")) { src = h->z4.p; len = h->l; } else if (!strcmp(name, "Ex")) { src = h->Ex.p; len = h->l; }
    else if (!strcmp(name, "live")) {
        if (cap < len) return lpbox_fail(LPBOX_E_BADARG, "buffer too small");
        std::vector<uint8_t> lv((size_t)len);
        HIPCHK(hipMemcpy(lv.data(), h->live.p, (size_t)len, hipMemcpyDeviceToHost));
        for (long j = 0; j < len; j++) out[j] = lv[j] ? 1.0 : 0.0;
        return (int)len;
    }
    else return lpbox_fail(LPBOX_E_BADARG, "unknown vector '%s'", name);
    if (cap < len) return lpbox_fail(LPBOX_E_BADARG, "buffer too small");
    HIPCHK(hipMemcpy(out, src, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    return (int)len;
}

int lpbox_big_check_infeasible(lpbox_big_t *h, int which) {
    if (!h || !h->inited) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    if (h->world != 1) return lpbox_fail(LPBOX_E_UNSUPPORTED, "the infeasibility counts need every column: one rank only");
    CHK(use_device(h));
    const int n = h->n_loc, l = h->l;
    std::vector<double> v(n);
    std::vector<uint8_t> live(n, 1);
    if (which == 0) {                                            // LPcpp:1577-1591: current E (live columns), raw iterate
        if (h->n_live_glob == 0) return 0;
        HIPCHK(hipMemcpy(v.data(), h->x.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(live.data(), h->live.p, (size_t)n, hipMemcpyDeviceToHost));
    } else {                                                     // LPcpp:1593-1612: original E, binary full-length solution
        int rc = lpbox_big_get_x_sol(h, v.data());
        if (rc < 0) return rc;
    }
    // rows in ascending column order (the slices of the row storage follow each other in column order)
    std::vector<double> s(l, 0.0);
    for (int ph = 0; ph < h->P; ph++)
        for (int i = 0; i < l; i++)
            for (int k = h->rptr[(size_t)ph * l + i]; k < h->rptr[(size_t)ph * l + i + 1]; k++)
                if (live[h->rcol[k]]) s[i] += 1.0 * v[h->rcol[k]];
    int inf = 0;
    for (int i = 0; i < l; i++) if (!(s[i] <= 1.0)) inf++;          // the reference compares with 1.0, not with f
    return inf;
}

int lpbox_big_get_scalar(lpbox_big_t *h, const char *name, double *out) {
    if (!h || !h->inited || !out || !name) return lpbox_fail(LPBOX_E_STATE, "not initialised");
    const BigState &s = h->hst;
    struct { const char *n; double v; } tab[] = {
        {"rho1", s.rho1}, {"rho4", s.rho4}, {"gamma", s.gamma_val}, {"dI", s.dI}, {"rho4Et", s.r4Et}, {"std_obj", s.std_obj},
        {"cur_obj", s.cur_obj}, {"best_bin_obj", s.best_bin_obj}, {"cvg1", s.cvg1}, {"cvg2", s.cvg2}, {"obj_val", s.obj_val},
        {"iter", (double)s.iter}, {"outer_total", (double)s.outer_total}, {"pcg_total", (double)s.pcg_total}, {"last_pcg", (double)s.last_pcg},
        {"sum_fix_obj", s.sum_fix_obj}, {"fix_obj", s.fix_obj}, {"c1", s.c1}, {"ret", (double)s.ret}, {"n_live", (double)h->n_live_glob},
        {"stop", (double)s.stop}, {"plain_iter_p1", (double)s.plain_iter_p1}, {"kmax", (double)h->kmax},
        {"launches", (double)h->launches}, {"collectives", (double)h->collectives}, {"kernel_ms", h->kernel_ms},
        {"threads", (double)BIG_T}, {"chunk", (double)(BIG_T * h->EPT)}, {"groups", (double)h->G}, {"row_slices", (double)h->P}, {"folded_reductions", h->fold ? 1.0 : 0.0}, {"pcg_comm_lean", h->lean ? 1.0 : 0.0},
        {"row_chunk", (double)((h->world > 1 || h->comm) ? BIG_T : (h->P > 1 ? BIG_T : BIG_T * h->EPTl))},
    };
    for (auto &e : tab) if (!strcmp(e.n, name)) { *out = e.v; return LPBOX_OK; }
    return lpbox_fail(LPBOX_E_BADARG, "unknown scalar '%s'", name);
}

}  // extern "C"
