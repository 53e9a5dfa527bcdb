// lpbox_capi.hip -- the C-ABI of liblpbox_hip.so (include/lpbox_hip.h): handle management, host-side index
// bookkeeping of early fixing, instance readers and result getters.  All solver arithmetic runs in the HIP kernels of
// lpbox_lp_kernels.hip; there is no CPU fallback.
//
// Reference citations: LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp,
//                      LPh   = .../cython_solver/LPboxADMMsolver.h, pxd = .../cython_solver/LPboxADMMsolver.pxd
#include "../../include/lpbox_hip.h"
#include "lpbox_lp.h"

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

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(LPBOX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct LpInstance {
    int n = 0, l = 0, nnz = 0;
    std::vector<int> colptr, rowidx;   // CSC of E, as read (LPcpp:2416-2444)
    std::vector<int> rowptr, colidx;   // CSR of the same matrix
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
    std::vector<LpInstance> inst;
    bool finalized = false, inited = false;
    int NS = 0, LS = 0, ZS = 0, T = 0, EPT = 0;
    size_t lds = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double kernel_ms = 0.0;
    long long launches = 0;
    DevBuf<int> csr_ptr, csc_ptr, isc, ctl, left_idx, xi_rows;
    DevBuf<uint16_t> csr_col, csc_row;
    DevBuf<double> x, z1, z2, b, pd, z4, f, f_org, dsc, hist, dctl, c1_init, xhist, xi_out;
    DevBuf<uint8_t> live, newfix;
    int ws_cap = 0;        // columns of the current xhist staging buffer
    int last_ws = 0;       // window length of the last l2f call
    bool xi_valid = false;
    long xi_out_stride = 0;
    int xi_out_ws = 0;
    std::vector<int> h_isc;   // host mirror, refreshed after every solver call
    std::vector<double> h_dsc;

    LpBatchDev dev() const {
        LpBatchDev d;
        d.B = B; d.NS = NS; d.LS = LS; d.ZS = ZS;
        d.csr_ptr = csr_ptr.p; d.csr_col = csr_col.p; d.csc_ptr = csc_ptr.p; d.csc_row = csc_row.p;
        d.x = x.p; d.z1 = z1.p; d.z2 = z2.p; d.b = b.p; d.pd = pd.p; d.live = live.p; d.newfix = newfix.p;
        d.z4 = z4.p; d.f = f.p; d.dsc = dsc.p; d.isc = isc.p; d.hist = hist.p;
        d.ctl = ctl.p; d.dctl = dctl.p; d.xhist = xhist.p; d.ws_cap = ws_cap;
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
    if (nmax > 65535 || lmax > 65535) return fail(LPBOX_E_UNSUPPORTED, "n or l exceeds the uint16 index range of the on-chip kernel");
    h->NS = (nmax + 7) & ~7; h->LS = (lmax + 7) & ~7; h->ZS = (zmax + 7) & ~7;
    int T = 512;
    if (const char *e = getenv("LPBOX_LP_THREADS")) { int v = atoi(e); if (v == 256 || v == 512 || v == 1024) T = v; }
    const int big = std::max(nmax, lmax);
    int EPT = (big + T - 1) / T;
    EPT = EPT <= 1 ? 1 : (EPT <= 2 ? 2 : 4);
    if ((long)T * EPT < big && T < 1024) { T = 1024; EPT = (big + T - 1) / T; EPT = EPT <= 1 ? 1 : (EPT <= 2 ? 2 : 4); }
    if ((long)T * EPT < big)
        return fail(LPBOX_E_UNSUPPORTED, "instance with max(n,l)=%d exceeds the on-chip kernel's %d register slots", big, T * EPT);
    h->T = T; h->EPT = EPT;
    h->lds = lp_window_lds_bytes(T, h->NS, h->LS, h->ZS);
    if (h->lds > 160 * 1024) return fail(LPBOX_E_UNSUPPORTED, "instance needs %zu B of LDS (> 160 KiB per CU)", h->lds);

    if (!h->stream) HIPCHK(hipStreamCreate(&h->stream));
    if (!h->ev0) { HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1)); }
    const size_t B = h->B, NS = h->NS, LS = h->LS, ZS = h->ZS;
    HIPCHK(h->csr_ptr.alloc(B * (LS + 1))); HIPCHK(h->csc_ptr.alloc(B * (NS + 1)));
    HIPCHK(h->csr_col.alloc(B * ZS)); HIPCHK(h->csc_row.alloc(B * ZS));
    HIPCHK(h->x.alloc(B * NS)); HIPCHK(h->z1.alloc(B * NS)); HIPCHK(h->z2.alloc(B * NS));
    HIPCHK(h->b.alloc(B * NS)); HIPCHK(h->pd.alloc(B * NS));
    HIPCHK(h->live.alloc(B * NS)); HIPCHK(h->newfix.alloc(B * NS));
    HIPCHK(h->z4.alloc(B * LS)); HIPCHK(h->f.alloc(B * LS)); HIPCHK(h->f_org.alloc(B * LS));
    HIPCHK(h->dsc.alloc(B * ND_COUNT)); HIPCHK(h->isc.alloc(B * NI_COUNT)); HIPCHK(h->hist.alloc(B * LP_HIST));
    HIPCHK(h->ctl.alloc(B * 4)); HIPCHK(h->dctl.alloc(B)); HIPCHK(h->c1_init.alloc(B));
    HIPCHK(h->left_idx.alloc(B * NS)); HIPCHK(h->xi_rows.alloc(B));

    std::vector<int> h_csr_ptr(B * (LS + 1), 0), h_csc_ptr(B * (NS + 1), 0), h_isc(B * NI_COUNT, 0);
    std::vector<uint16_t> h_csr_col(B * ZS, 0), h_csc_row(B * ZS, 0);
    std::vector<double> h_b(B * NS, 0.0), h_f(B * LS, 0.0), h_c1(B, 0.0);
    for (size_t i = 0; i < B; i++) {
        const LpInstance &I = h->inst[i];
        for (int r = 0; r <= I.l; r++) h_csr_ptr[i * (LS + 1) + r] = I.rowptr[r];
        for (int c = 0; c <= I.n; c++) h_csc_ptr[i * (NS + 1) + c] = I.colptr[c];
        for (int k = 0; k < I.nnz; k++) { h_csr_col[i * ZS + k] = (uint16_t)I.colidx[k]; h_csc_row[i * ZS + k] = (uint16_t)I.rowidx[k]; }
        for (int j = 0; j < I.n; j++) h_b[i * NS + j] = I.b[j];
        for (int r = 0; r < I.l; r++) h_f[i * LS + r] = I.f_org[r];
        h_isc[i * NI_COUNT + NI_N] = I.n; h_isc[i * NI_COUNT + NI_L] = I.l; h_isc[i * NI_COUNT + NI_NNZ] = I.nnz;
        h_isc[i * NI_COUNT + NI_ACTIVE] = 1;
        h_c1[i] = std::pow((double)I.n, 1.0 / 2);     // std::pow(n, 1.0/p), p = projection_lp = 2 (LPcpp:427,503)
    }
    HIPCHK(hipMemcpy(h->csr_ptr.p, h_csr_ptr.data(), h_csr_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->csc_ptr.p, h_csc_ptr.data(), h_csc_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->csr_col.p, h_csr_col.data(), h_csr_col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->csc_row.p, h_csc_row.data(), h_csc_row.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
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

int run_window(lpbox_t *h, int iter_start, int iter_end, int l2f) {
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(lp_launch_window(h->dev(), h->T, h->EPT, h->lds, iter_start, iter_end, l2f, h->stream));
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
    LpInstance &I = h->inst[idx];
    I.n = n; I.l = l; I.nnz = nnz;
    I.colptr.assign(colptr, colptr + n + 1);
    I.rowidx.assign(rowidx, rowidx + nnz);
    for (int j = 0; j < n; j++) {
        if (colptr[j + 1] < colptr[j]) return fail(LPBOX_E_BADARG, "colptr not monotone");
        for (int k = colptr[j]; k < colptr[j + 1]; k++) {
            if (rowidx[k] < 0 || rowidx[k] >= l) return fail(LPBOX_E_BADARG, "row index out of range");
            if (k > colptr[j] && rowidx[k] <= rowidx[k - 1]) return fail(LPBOX_E_BADARG, "row indices must ascend inside a column");
            if (vals && vals[k] != 1.0)
                return fail(LPBOX_E_UNSUPPORTED, "E has a stored value %g != 1; the LP kernels hold E implicitly as a 0/1 pattern", vals[k]);
        }
    }
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

int fetch_vec(lpbox_t *h, const double *pool, size_t stride, int idx, int len, std::vector<double> &out) {
    out.resize(len);
    if (len == 0) return LPBOX_OK;
    HIPCHK(hipMemcpy(out.data(), pool + (size_t)idx * stride, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    return LPBOX_OK;
}

int fetch_live(lpbox_t *h, int idx, std::vector<uint8_t> &out) {
    out.resize(h->inst[idx].n);
    HIPCHK(hipMemcpy(out.data(), h->live.p + (size_t)idx * h->NS, out.size(), hipMemcpyDeviceToHost));
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
    if (batch <= 0 || (flavour != LPBOX_FLAVOUR_LP)) {
        fail(LPBOX_E_BADARG, "lpbox_create: unsupported flavour %d or batch %d", flavour, batch);
        return nullptr;
    }
    lpbox_t *h = new lpbox_solver();
    h->flavour = flavour; h->B = batch; h->print_info = print_info; h->device = g_device;
    h->inst.resize(batch);
    return h;
}

void lpbox_destroy(lpbox_t *h) {
    if (!h) return;
    if (h->finalized) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->csr_ptr.release(); h->csc_ptr.release(); h->isc.release(); h->ctl.release(); h->left_idx.release(); h->xi_rows.release();
    h->csr_col.release(); h->csc_row.release();
    h->x.release(); h->z1.release(); h->z2.release(); h->b.release(); h->pd.release(); h->z4.release(); h->f.release();
    h->f_org.release(); h->dsc.release(); h->hist.release(); h->dctl.release(); h->c1_init.release(); h->xhist.release();
    h->xi_out.release(); h->live.release(); h->newfix.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int lpbox_set_problem_lp(lpbox_t *h, int idx, int n, int l, int nnz, const int *colptr, const int *rowidx,
                         const double *vals, const double *b, const double *f) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    return set_instance(h, idx, n, l, nnz, colptr, rowidx, vals, b, f);
}

// readSparseMat LPcpp:2416-2444, readDenseVec :2407-2414, readFile :2446-2545
int lpbox_read_files_lp(lpbox_t *h, int idx, const char *path_C, const char *path_b, int k) {
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
    HIPCHK(lp_launch_init(h->dev(), h->T, h->EPT, h->f_org.p, h->c1_init.p, h->stream));
    rc = refresh_scalars(h);
    if (rc) return rc;
    h->inited = true;
    return 1;                                                              // LPcpp:762
}

int lpbox_iterate(lpbox_t *h, int iter_start, int iter_end, int *rets) {
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (!h->inited) return fail(LPBOX_E_STATE, "solve_init has not been called");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(h->ctl.p, 0, (size_t)h->B * 4 * sizeof(int), h->stream));
    rc = run_window(h, iter_start, iter_end, 0);
    if (rc) return rc;
    for (int i = 0; i < h->B; i++)
        if (rets) rets[i] = h->h_isc[(size_t)i * NI_COUNT + NI_RET];
    return h->h_isc[NI_RET];
}

int lpbox_iterate_l2f(lpbox_t *h, int iter_start, int iter_end, const double *vec, long vec_stride, const int *nums,
                      int *rets) {
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
        if (num < 0 || num > n_live) return fail(LPBOX_E_BADARG, "instance %zu: fix count %d outside [0,%d]", i, num, n_live);
        if (num != 0) {
            if (!vec) return fail(LPBOX_E_BADARG, "fix vector missing");
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
                if (v[q] == 1) h_newfix[i * NS + org] = 2;
                else if (v[q] == 0) h_newfix[i * NS + org] = 1;
                else keep.push_back(org);
            }
            I.left_idx.swap(keep);
            h_ctl[i * 4 + 0] = 1; h_ctl[i * 4 + 1] = num; h_ctl[i * 4 + 2] = n_live - num;
            h_dctl[i] = std::pow((double)(n_live - num), 1.0 / 2);         // LPcpp:427 with the shrunken n
        }
        I.xi_rows = n_live - num;                                           // x_iters = Zero(n - fix_num, 500), :1113
        I.xi_left_idx = I.left_idx;
        h_rows[i] = I.xi_rows;
        for (int q = 0; q < I.xi_rows; q++) h_left[i * NS + q] = I.left_idx[q];
    }
    if (ws > 0 && (h->ws_cap < ws || !h->xhist.p)) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(h->xhist.alloc(B * (size_t)ws * NS));
        h->ws_cap = ws;
    }
    HIPCHK(hipMemcpyAsync(h->ctl.p, h_ctl.data(), h_ctl.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->dctl.p, h_dctl.data(), h_dctl.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (any_fix) HIPCHK(hipMemcpyAsync(h->newfix.p, h_newfix.data(), h_newfix.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->left_idx.p, h_left.data(), h_left.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->xi_rows.p, h_rows.data(), h_rows.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (ws > 0) HIPCHK(hipMemsetAsync(h->xhist.p, 0, B * (size_t)h->ws_cap * NS * sizeof(double), h->stream));   // x_iters starts as zeros
    rc = run_window(h, iter_start, iter_end, 1);    // synchronises: the host staging vectors above stay alive until here
    if (rc) return rc;
    h->last_ws = ws;
    h->xi_valid = true;
    h->xi_out_ws = 0;
    for (size_t i = 0; i < B; i++) {
        if (rets) rets[i] = h->h_isc[i * NI_COUNT + NI_RET];
    }
    return h->h_isc[NI_RET];
}

int lpbox_get_n(lpbox_t *h, int idx) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return h->inst[idx].n;
    return h->h_isc[(size_t)idx * NI_COUNT + NI_NLIVE];
}

int lpbox_get_org_n(lpbox_t *h, int idx) {
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
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return 0;
    return h->h_isc[(size_t)idx * NI_COUNT + NI_ITER];
}

int lpbox_get_x_iters(lpbox_t *h, int idx, int ws, double *out) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->xi_valid) return fail(LPBOX_E_STATE, "solve_iter_l2f has not been called");
    if (ws < 0 || ws > LP_XITERS_COLS) return fail(LPBOX_E_BADARG, "ws = %d outside [0,%d]", ws, LP_XITERS_COLS);
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

int lpbox_get_x_sol(lpbox_t *h, int idx, double *out) {
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
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    int rc = finalize(h);
    if (rc) return rc;
    if (threads) *threads = h->T;
    if (elems_per_thread) *elems_per_thread = h->EPT;
    if (lds_bytes) *lds_bytes = (int)h->lds;
    return LPBOX_OK;
}

int lpbox_get_counters(lpbox_t *h, int idx, long long *outer_iters, long long *pcg_iters) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return fail(LPBOX_E_STATE, "not initialised");
    if (outer_iters) *outer_iters = h->h_isc[(size_t)idx * NI_COUNT + NI_OUTER_TOTAL];
    if (pcg_iters) *pcg_iters = h->h_isc[(size_t)idx * NI_COUNT + NI_PCG_TOTAL];
    return LPBOX_OK;
}

int lpbox_get_stop(lpbox_t *h, int idx, int *reason, int *plain_iter_plus1) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited) return fail(LPBOX_E_STATE, "not initialised");
    if (reason) *reason = h->h_isc[(size_t)idx * NI_COUNT + NI_STOP];
    if (plain_iter_plus1) *plain_iter_plus1 = h->h_isc[(size_t)idx * NI_COUNT + NI_PLAIN_ITER_P1];
    return LPBOX_OK;
}

int lpbox_kernel_time(lpbox_t *h, double *ms_total, long long *launches, int reset) {
    if (!valid_handle(h)) return fail(LPBOX_E_BADHANDLE, "bad handle");
    if (ms_total) *ms_total = h->kernel_ms;
    if (launches) *launches = h->launches;
    if (reset) { h->kernel_ms = 0.0; h->launches = 0; }
    return LPBOX_OK;
}

int lpbox_debug_get_vec(lpbox_t *h, int idx, const char *name, double *out, int cap) {
    int rc = check_idx(h, idx);
    if (rc) return rc;
    if (!h->inited || !name || !out) return fail(LPBOX_E_STATE, "not initialised");
    rc = use_device(h);
    if (rc) return rc;
    const LpInstance &I = h->inst[idx];
    const double *pool = nullptr; size_t stride = h->NS; int len = I.n;
    if (!strcmp(name, "x")) pool = h->x.p;
    else if (!strcmp(name, "z1")) pool = h->z1.p;
    else if (!strcmp(name, "z2")) pool = h->z2.p;
    else if (!strcmp(name, "b")) pool = h->b.p;
    else if (!strcmp(name, "pd")) pool = h->pd.p;
    else if (!strcmp(name, "z4")) { pool = h->z4.p; stride = h->LS; len = I.l; }
    else if (!strcmp(name, "f")) { pool = h->f.p; stride = h->LS; len = I.l; }
    else if (!strcmp(name, "live")) {
        std::vector<uint8_t> live;
        if ((rc = fetch_live(h, idx, live))) return rc;
        if (I.n > cap) return fail(LPBOX_E_BADARG, "buffer too small");
        for (int j = 0; j < I.n; j++) out[j] = live[j];
        return I.n;
    } else return fail(LPBOX_E_BADARG, "unknown vector '%s'", name);
    if (len > cap) return fail(LPBOX_E_BADARG, "buffer too small");
    std::vector<double> v;
    if ((rc = fetch_vec(h, pool, stride, idx, len, v))) return rc;
    memcpy(out, v.data(), sizeof(double) * (size_t)len);
    return len;
}

int lpbox_debug_get_scalar(lpbox_t *h, int idx, const char *name, double *out) {
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

}  // extern "C"
