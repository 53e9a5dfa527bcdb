// LPboxADMMsolver.cpp -- bodies of the host-side class declared in LPboxADMMsolver.h: every method forwards to the C-ABI of
// liblpbox_hip.so (include/lpbox_hip.h).  Build: g++ -I<repo>/include ... -L<repo>/accelerated-lpbox-admm_amd/lpbox_hip -llpbox_hip.
// The reference's pxd includes this file textually (LPboxADMMsolver.pxd:1-2), so everything here is `inline` or in an unnamed
// namespace and the file may also be compiled on its own.
#ifndef LPBOX_ADMM_SOLVER_CPP_INCLUDED
#define LPBOX_ADMM_SOLVER_CPP_INCLUDED
#include "LPboxADMMsolver.h"

#include <lpbox_hip.h>

#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace {
constexpr int LPBOX_ONCHIP_MAX = 2048;     // storage positions of the one-workgroup-per-instance kernel (512 threads x 4 slots)

inline bool lpbox_quiet() { const char *e = getenv("LPBOX_QUIET"); return e && *e && *e != '0'; }
inline void lpbox_throw(const char *what) {
    const char *m = lpbox_last_error();
    throw std::runtime_error(std::string(what) + " failed: " + (m ? m : ""));
}
inline int lpbox_ok(int rc, const char *what) { if (rc < 0) lpbox_throw(what); return rc; }
inline bool lpbox_is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
}  // namespace

struct LPboxADMMsolver::State {
    lpbox_t *h = nullptr;            // on-chip path (one instance)
    lpbox_big_t *big = nullptr;      // large-instance path, one rank (set by ADMM_lp_iters_init when the instance does not fit a CU)
    int print_info = 0, consistency = 5;
    double fix_threshold = 1e-3;
    std::string root;                // "" = LPBOX_DATA_ROOT or the reference's "../cython_solver/data"
    bool have_file = false; int file_i = 0, file_k = 0, file_j = 0;
    int org_n = 0, l = 0;
    std::vector<double> xiters, xsol, xfinal, x_prev;
    bool does_log = false;
    ~State() { if (big) lpbox_big_destroy(big); if (h) lpbox_destroy(h); }
    std::string data_root() const {
        if (!root.empty()) return root;
        const char *e = getenv("LPBOX_DATA_ROOT");
        return (e && *e) ? std::string(e) : std::string("../cython_solver/data");       // LPcpp:2451
    }
    void fresh() {
        if (big) { lpbox_big_destroy(big); big = nullptr; }
        if (h) { lpbox_destroy(h); h = nullptr; }
        h = lpbox_create(LPBOX_FLAVOUR_LP, 1, print_info);
        if (!h) lpbox_throw("lpbox_create");
        x_prev.clear();
    }
    double scalar(const char *name) {
        double v = 0;
        if (big) lpbox_ok(lpbox_big_get_scalar(big, name, &v), "lpbox_big_get_scalar");
        else lpbox_ok(lpbox_debug_get_scalar(h, 0, name, &v), "lpbox_debug_get_scalar");
        return v;
    }
    int stop(int *p1) {
        int r = 0, p = 0;
        if (big) { r = (int)scalar("stop"); p = (int)scalar("plain_iter_p1"); }
        else lpbox_ok(lpbox_get_stop(h, 0, &r, &p), "lpbox_get_stop");
        if (p1) *p1 = p;
        return r;
    }
    int n_live() { return big ? lpbox_ok(lpbox_big_get_n(big), "lpbox_big_get_n") : lpbox_ok(lpbox_get_n(h, 0), "lpbox_get_n"); }
    int iter() { return big ? (int)scalar("iter") : lpbox_ok(lpbox_get_iter(h, 0), "lpbox_get_iter"); }
    void echo_stop(bool plain) {                       // the reference's stdout lines (LPcpp:935, :984 / :1506, :1540)
        if (lpbox_quiet()) return;
        int p1 = 0;
        const int reason = stop(&p1);
        const int it = plain ? p1 - 1 : iter();
        if (reason == 1)
            printf(plain ? "Stop because y1_y2. iter: %d, stop_threshold: %.6f\n" : "Stop becuase y1_y2. iter: %d, stop_threshold: %.6f\n", it,
                   std::max(scalar("cvg1"), scalar("cvg2")));
        else if (reason == 2)
            printf(plain ? "Stop because obj_std. iter: %d, std_threshold: %.6f\n" : "Stop because std_obj. iter: %d, std_threshold: %.6f\n", it,
                   scalar("std_obj"));
    }
};

inline LPboxADMMsolver::LPboxADMMsolver() : s_(std::make_shared<State>()) { s_->fresh(); }
inline LPboxADMMsolver::LPboxADMMsolver(int print_info) : s_(std::make_shared<State>()) {
    if (!lpbox_quiet()) printf("Object with fix_info is created!\n");                  // LPcpp:478
    s_->print_info = print_info;
    s_->fresh();
}
inline LPboxADMMsolver::LPboxADMMsolver(int consistency, double fix_threshold) : s_(std::make_shared<State>()) {
    s_->consistency = consistency; s_->fix_threshold = fix_threshold;
    s_->fresh();
}
inline void LPboxADMMsolver::set_fix_threshold(double t) { s_->fix_threshold = t; }
inline void LPboxADMMsolver::set_consistency(int c) { s_->consistency = c; }
inline void LPboxADMMsolver::set_data_root(const std::string &root) { s_->root = root; }
inline bool LPboxADMMsolver::on_large_path() const { return s_->big != nullptr; }
inline int LPboxADMMsolver::get_org_n() { return s_->org_n; }

inline void LPboxADMMsolver::readFile(int i, int k, int j) {
    s_->fresh();
    lpbox_ok(lpbox_read_file(s_->h, 0, s_->data_root().c_str(), i, k, j), "lpbox_read_file");
    s_->have_file = true; s_->file_i = i; s_->file_k = k; s_->file_j = j;
}

inline void LPboxADMMsolver::set_problem(int n, int l, const int *colptr, const int *rowidx, const double *b, const double *f) {
    s_->fresh();
    lpbox_ok(lpbox_set_problem_lp(s_->h, 0, n, l, colptr ? colptr[n] : 0, colptr, rowidx, nullptr, b, f), "lpbox_set_problem_lp");
    s_->have_file = false;
}

inline int LPboxADMMsolver::ADMM_lp_iters_init() {
    State &s = *s_;
    s.x_prev.clear();                                      // x_prev = Zero(n) (LPcpp:572)
    if (!s.big) {
        int n = 0, l = 0, nnz = 0;
        lpbox_ok(lpbox_get_problem_lp(s.h, 0, &n, &l, &nnz, nullptr, nullptr, nullptr, nullptr), "lpbox_get_problem_lp");
        s.org_n = n; s.l = l;
        bool fits = std::max(n, l) <= LPBOX_ONCHIP_MAX;
        if (fits) {
            const int rc = lpbox_init(s.h);
            if (rc >= 0) return rc;
            if (rc != LPBOX_E_TOOLARGE) lpbox_throw("lpbox_init");
        }
        // does not fit one CU: the same algorithm on the large-instance path, one rank
        std::vector<int> colptr((size_t)n + 1), rowidx((size_t)std::max(nnz, 1));
        std::vector<double> b((size_t)n), f((size_t)l);
        lpbox_ok(lpbox_get_problem_lp(s.h, 0, nullptr, nullptr, nullptr, colptr.data(), rowidx.data(), b.data(), f.data()), "lpbox_get_problem_lp");
        s.big = lpbox_big_create(0, 1, 0);
        if (!s.big) lpbox_throw("lpbox_big_create");
        lpbox_ok(lpbox_big_set_problem(s.big, n, 0, n, l, colptr.data(), rowidx.data(), b.data(), f.data()), "lpbox_big_set_problem");
        lpbox_destroy(s.h); s.h = nullptr;
    }
    return lpbox_ok(lpbox_big_init(s.big), "lpbox_big_init");
}

inline int LPboxADMMsolver::ADMM_lp_iters(int iter_start, int iter_end) {
    State &s = *s_;
    // side-effect files (LPcpp:776-783, :903-909, :1081): <root>/xiter/allres.csv gets one line per call; print_info 2 / 3 dump the
    // iterates (both paths; print_info 3 on the large-instance route reads the final iterate back).  Written iff <root>/xiter exists
    // (the reference crashes without it).  does_log: <root>/log/<k>_<j>_log_<i>.txt iff asked for and <root>/log exists.
    const std::string xdir = s.data_root() + "/xiter";
    const bool files = s.have_file && lpbox_is_dir(xdir);
    const bool dump = files && (s.print_info == 2 || s.print_info == 3) && iter_end > iter_start;
    // print_info 3 writes only the iterate of the stop (LPcpp:940-946): on the large-instance route it is read back after the solve
    const bool final_only = dump && s.big && s.print_info == 3;
    if (!s.big) lpbox_ok(lpbox_set_record(s.h, dump ? 1 : 0), "lpbox_set_record");
    else lpbox_ok(lpbox_big_set_record(s.big, dump && !final_only ? 1 : 0), "lpbox_big_set_record");
    const std::string ldir = s.data_root() + "/log";
    const bool log = s.does_log && !s.big && s.have_file && iter_end > iter_start && lpbox_is_dir(ldir);
    if (!s.big) lpbox_ok(lpbox_set_log(s.h, log ? 1 : 0), "lpbox_set_log");
    const auto t0 = std::chrono::steady_clock::now();
    int ret = 0;
    if (s.big) lpbox_ok(lpbox_big_iterate(s.big, iter_start, iter_end, &ret), "lpbox_big_iterate");
    else lpbox_ok(lpbox_iterate(s.h, iter_start, iter_end, &ret), "lpbox_iterate");
    const double secs = 1.0 * (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000;
    s.echo_stop(true);
    if (files) {
        int p1 = 0;
        const int reason = s.stop(&p1);
        if (final_only) {
            if (reason == 1 || reason == 2) {
                char name[64];
                snprintf(name, sizeof name, "/%d_%d_xiters_%d.csv", s.file_k, s.file_j, s.file_i);
                if (FILE *xi = fopen((xdir + name).c_str(), "w+")) {
                    const double *x = get_final_x_sol();
                    const int rows = get_n();
                    fprintf(xi, "Iter%d,", p1);
                    for (int r = 0; r < rows; r++) fprintf(xi, r + 1 < rows ? "%lf," : "%lf", x[r]);
                    fprintf(xi, "\n");
                    fclose(xi);
                }
            }
        } else if (dump) {
            const int ws = iter_end - iter_start;
            const int done = (reason == 1 || reason == 2) ? p1 - iter_start : ws;
            const int rows = s.big ? lpbox_ok(lpbox_big_get_x_iters(s.big, ws, nullptr), "lpbox_big_get_x_iters")
                                   : lpbox_ok(lpbox_get_x_iters(s.h, 0, ws, nullptr), "lpbox_get_x_iters");
            std::vector<double> X((size_t)rows * ws);
            if (rows) {
                if (s.big) lpbox_ok(lpbox_big_get_x_iters(s.big, ws, X.data()), "lpbox_big_get_x_iters");
                else lpbox_ok(lpbox_get_x_iters(s.h, 0, ws, X.data()), "lpbox_get_x_iters");
            }
            const int lo = s.print_info == 2 ? 0 : ((reason == 1 || reason == 2) ? done - 1 : done);
            char name[64];
            snprintf(name, sizeof name, "/%d_%d_xiters_%d.csv", s.file_k, s.file_j, s.file_i);
            if (FILE *xi = fopen((xdir + name).c_str(), "w+")) {
                for (int c = lo; c < done; c++) {
                    fprintf(xi, "Iter%d,", iter_start + c + 1);
                    for (int r = 0; r < rows; r++) fprintf(xi, r + 1 < rows ? "%lf," : "%lf", X[(size_t)r * ws + c]);
                    fprintf(xi, "\n");
                }
                fclose(xi);
            }
        }
        if (FILE *al = fopen((xdir + "/allres.csv").c_str(), "a")) {
            fprintf(al, "%d,%f,%d,%f\n", s.file_i, -get_curBinObj(), p1, secs);          // LPcpp:1081
            fclose(al);
        }
    }
    if (log) {                                                       // LPcpp:772 ("w+"), :789, :898-901, :1013-1067
        int p1 = 0;
        const int reason = s.stop(&p1);
        const int ws = iter_end - iter_start;
        std::vector<double> rec((size_t)ws * LPBOX_LOG_VALS);
        const int rows = lpbox_ok(lpbox_get_log(s.h, 0, rec.data(), ws), "lpbox_get_log");
        char name[64];
        snprintf(name, sizeof name, "/%d_%d_log_%d.txt", s.file_k, s.file_j, s.file_i);
        if (FILE *fp = fopen((ldir + name).c_str(), "w+")) {
            for (int r = 0; r < rows; r++) {
                const double *v = &rec[(size_t)r * LPBOX_LOG_VALS];
                fprintf(fp, "Iteration: %d\n", (int)v[11]);
                fprintf(fp, "Conjugate gradient stops after %d iterations\n", (int)v[0]);
                fprintf(fp, "norm of x_sol: %.9lf\nnorm of y1: %.9lf\nnorm of y2: %.9lf\nnorm of y3: %.9lf\n", v[1], v[2], v[3], v[4]);
                fprintf(fp, "norm of z1: %.9lf\nnorm of z2: %.9lf\nFor z4\nnorm of z4: %.9lf\n", v[5], v[6], v[7]);
                fprintf(fp, "LongkangIter: %d;  x_sol: %lf; dou_obj:%lf; bin_obj: %lf\n", (int)v[11] + 1, v[1], v[8], v[9]);
                fprintf(fp, "Time elapsed: %lfs\n", v[10]);
                fprintf(fp, "-------------------------------------------------\n");
            }
            if (reason == 1 || reason == 2) fprintf(fp, "Iteration: %d\n", p1 - 1);
            fclose(fp);
        }
    }
    return ret;
}

inline void LPboxADMMsolver::set_does_log(int on) { s_->does_log = on != 0; }

inline int LPboxADMMsolver::ADMM_lp_iters_l2f(int iter_start, int iter_end, double *vec, int num) {
    State &s = *s_;
    int ret = 0;
    if (s.big) lpbox_ok(lpbox_big_iterate_l2f(s.big, iter_start, iter_end, vec, num, &ret), "lpbox_big_iterate_l2f");
    else {
        const int n_live = s.n_live();
        lpbox_ok(lpbox_iterate_l2f(s.h, iter_start, iter_end, num ? vec : nullptr, n_live, num ? &num : nullptr, &ret), "lpbox_iterate_l2f");
    }
    s.echo_stop(false);
    return ret;
}

inline double LPboxADMMsolver::cal_obj() {
    double v = 0;
    if (s_->big) lpbox_ok(lpbox_big_cal_obj(s_->big, &v), "lpbox_big_cal_obj"); else lpbox_ok(lpbox_cal_obj(s_->h, 0, &v), "lpbox_cal_obj");
    return v;
}
inline double LPboxADMMsolver::get_curBinObj() {
    double v = 0;
    if (s_->big) return s_->scalar("cur_obj");
    lpbox_ok(lpbox_cur_bin_obj(s_->h, 0, &v), "lpbox_cur_bin_obj");
    return v;
}
inline int LPboxADMMsolver::get_n() { return s_->n_live(); }
inline int LPboxADMMsolver::get_iter() { return s_->iter(); }

inline double *LPboxADMMsolver::get_x_iters_d(int ws) {
    State &s = *s_;
    const int rows = s.big ? lpbox_ok(lpbox_big_get_x_iters(s.big, ws, nullptr), "lpbox_big_get_x_iters")
                           : lpbox_ok(lpbox_get_x_iters(s.h, 0, ws, nullptr), "lpbox_get_x_iters");
    s.xiters.assign((size_t)std::max(rows, 1) * std::max(ws, 1), 0.0);
    if (rows && ws) {
        if (s.big) lpbox_ok(lpbox_big_get_x_iters(s.big, ws, s.xiters.data()), "lpbox_big_get_x_iters");
        else lpbox_ok(lpbox_get_x_iters(s.h, 0, ws, s.xiters.data()), "lpbox_get_x_iters");
    }
    return s.xiters.data();
}

inline double *LPboxADMMsolver::get_x_sol() {
    State &s = *s_;
    s.xsol.assign((size_t)std::max(s.org_n, 1), 0.0);
    if (s.big) lpbox_ok(lpbox_big_get_x_sol(s.big, s.xsol.data()), "lpbox_big_get_x_sol");
    else lpbox_ok(lpbox_get_x_sol(s.h, 0, s.xsol.data()), "lpbox_get_x_sol");
    return s.xsol.data();
}

inline double *LPboxADMMsolver::get_final_x_sol() {
    State &s = *s_;
    s.xfinal.assign((size_t)std::max(s.org_n, 1), 0.0);
    if (s.big) {
        std::vector<double> x((size_t)s.org_n), live((size_t)std::max(s.org_n, s.l));
        lpbox_ok(lpbox_big_get_x(s.big, x.data()), "lpbox_big_get_x");
        lpbox_ok(lpbox_big_get_vec(s.big, "live", live.data(), (long)live.size()), "lpbox_big_get_vec");
        size_t k = 0;
        for (int j = 0; j < s.org_n; j++) if (live[j] != 0) s.xfinal[k++] = x[j];
    } else lpbox_ok(lpbox_get_final_x_sol(s.h, 0, s.xfinal.data()), "lpbox_get_final_x_sol");
    return s.xfinal.data();
}

inline int LPboxADMMsolver::check_infeasible_lpbox() {
    State &s = *s_;
    const int inf = s.big ? lpbox_ok(lpbox_big_check_infeasible(s.big, 0), "lpbox_big_check_infeasible")
                          : lpbox_ok(lpbox_check_infeasible_lpbox(s.h, 0), "lpbox_check_infeasible_lpbox");
    if (!lpbox_quiet()) printf("Total constraints: [%d], Feasible: [%d], Infeasible: [%d]\n", s.l, s.l - inf, inf);   // LPcpp:1589
    return inf;
}
inline int LPboxADMMsolver::check_infeasible_l2f() {
    State &s = *s_;
    const int inf = s.big ? lpbox_ok(lpbox_big_check_infeasible(s.big, 1), "lpbox_big_check_infeasible")
                          : lpbox_ok(lpbox_check_infeasible_l2f(s.h, 0), "lpbox_check_infeasible_l2f");
    if (!lpbox_quiet()) printf("Total constraints: [%d], Feasible: [%d], Infeasible: [%d]\n", s.l, s.l - inf, inf);   // LPcpp:1610
    return inf;
}

inline long long LPboxADMMsolver::outer_iterations() {
    if (s_->big) return (long long)s_->scalar("outer_total");
    long long o = 0, p = 0;
    lpbox_ok(lpbox_get_counters(s_->h, 0, &o, &p), "lpbox_get_counters");
    return o;
}
inline int LPboxADMMsolver::stop_reason(int *plain_iter_plus1) { return s_->stop(plain_iter_plus1); }

// ADMM_lp_iters_fix (LPcpp:1689-2286) with the repaired semantics of DESIGN.md section 16 -- the loop of
// lpbox_hip/lp.py:PyLPboxADMMsolver.solve_iter_fix: one single-iteration l2f window per iteration (the rule needs every iterate
// on the host), a variable whose iterate moved by <= fix_threshold for `consistency` consecutive iterations is flagged
// (:1857-1871), more than 10 flagged variables are fixed at their rounded value at the start of the next iteration (:1932, :2006).
inline int LPboxADMMsolver::ADMM_lp_iters_fix(int iter_start, int iter_end) {
    State &s = *s_;
    int n_live = s.n_live();
    if ((int)s.x_prev.size() != n_live) s.x_prev.assign((size_t)n_live, 0.0);            // x_prev = Zero(n) (:572)
    std::vector<double> count((size_t)n_live, 0.0), vec;
    std::vector<char> flag((size_t)n_live, 0);
    int num = 0, ret = 0;
    auto compact = [&]() {                               // the counters follow their variable through the fix
        size_t k = 0;
        for (size_t j = 0; j < vec.size(); j++)
            if (vec[j] == -1) { s.x_prev[k] = s.x_prev[j]; count[k] = count[j]; flag[k] = flag[j]; k++; }
        s.x_prev.resize(k); count.resize(k); flag.resize(k);
        n_live = (int)k;
    };
    std::vector<double> zeros;
    for (int it = iter_start; it < iter_end; it++) {
        zeros.assign((size_t)std::max(n_live, 1), 0.0);
        const int r = ADMM_lp_iters_l2f(it, it + 1, num ? vec.data() : zeros.data(), num);
        if (num) { compact(); vec.clear(); num = 0; }
        const int reason = s.stop(nullptr);
        if (reason == 3 || reason == 4 || (r && reason == 0)) { ret = 1; break; }        // :1804-1807 / everything fixed / |x| < 1e-3 (:2000)
        const double *x = get_x_iters_d(1);
        for (int j = 0; j < n_live; j++) {
            const bool det = std::fabs(x[j] - s.x_prev[j]) <= s.fix_threshold;           // :1861
            count[j] = det ? count[j] + 1 : 0.0;
            if (det && count[j] >= s.consistency) flag[j] = 1;
            s.x_prev[j] = x[j];                                                           // :1871
        }
        if (reason == 1) break;                                                           // y1_y2: break, ret stays 0 (:1880-1886)
        if (reason == 2) { ret = 1; break; }                                              // obj_std (:1913-1919)
        int fix_n = 0;
        for (int j = 0; j < n_live; j++) fix_n += flag[j];
        if (fix_n > 10) {                                                                 // :1929-1932
            vec.assign((size_t)n_live, -1.0);
            for (int j = 0; j < n_live; j++) if (flag[j]) vec[j] = x[j] >= 0.5 ? 1.0 : 0.0;
            num = fix_n;
        }
    }
    if (num) {                                            // a fix decided by the last iteration goes in now (:1935), zero-length window
        const int r = ADMM_lp_iters_l2f(iter_end, iter_end, vec.data(), num);
        compact();
        ret = ret || r;
    }
    return ret;
}
#endif  // LPBOX_ADMM_SOLVER_CPP_INCLUDED
