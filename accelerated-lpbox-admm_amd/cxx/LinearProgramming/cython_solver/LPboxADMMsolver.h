// LPboxADMMsolver.h -- the reference's C++ solver class, LP flavour, as a thin host-side class over the C-ABI of liblpbox_hip.so.
//
// It has the public interface the reference's callers use -- LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.h
// :296-400 as bound by LPboxADMMsolver.pxd:5-21 and driven by test.cpp:10-18 -- with the same names, argument meaning and return
// values, so that test.cpp and the Cython module lpbox.pyx build against THIS pair of files (LPboxADMMsolver.h / .cpp) unchanged
// and run the MI355X kernels; see INTEGRATION.md section 3b.  There is no solver arithmetic in this class: every method forwards to
// include/lpbox_hip.h (the symbol is named next to each method).  Differences from the reference, all deliberate:
//   * copies share one solver (the Cython class holds the object by value and assigns a temporary to it, lpbox.pyx:8-14);
//   * the arrays behind get_x_iters_d / get_x_sol / get_final_x_sol belong to the object and stay valid until the next call of the
//     same getter (the reference `new`s one per call and nobody frees it);
//   * a failing call throws std::runtime_error with lpbox_last_error() (the pxd declares the constructors `except +`; the
//     reference dereferences NULL files or calls exit());
//   * an instance beyond the on-chip kernel (max(n, l) > 2048, or too dense for a CU's LDS) is handed to the large-instance
//     path by ADMM_lp_iters_init(), behind the same methods (the reference has no size limit);
//   * data root: the reference's "../cython_solver/data" relative to the CWD (LPcpp:2451) unless LPBOX_DATA_ROOT is set or
//     set_data_root() is called; LPBOX_QUIET=1 silences the stdout lines the reference prints.
#pragma once
#include <memory>
#include <string>
#include <vector>

class LPboxADMMsolver {
public:
    LPboxADMMsolver();                                        // LPh:298, LPcpp:472-475 (print_info 0)
    LPboxADMMsolver(int print_info);                          // LPh:300, LPcpp:477-480
    LPboxADMMsolver(int consistency, double fix_threshold);   // LPh:302, LPcpp:482-486: parameters of ADMM_lp_iters_fix
    void free() {}                                            // LPh:319 (the reference leaks; here the destructor of the last copy releases)

    void readFile(int i, int k, int j);                       // lpbox_read_file            LPcpp:2446-2545
    int ADMM_lp_iters_init();                                 // lpbox_init                 LPcpp:489-763
    int ADMM_lp_iters(int iter_start, int iter_end);          // lpbox_iterate              LPcpp:766-1095
    int ADMM_lp_iters_l2f(int iter_start, int iter_end, double *vec, int num);   // lpbox_iterate_l2f  LPcpp:1098-1574
    int ADMM_lp_iters_fix(int iter_start, int iter_end);      // host-driven on lpbox_iterate_l2f, repaired semantics, LPcpp:1689-2286
    double cal_obj();                                         // lpbox_cal_obj              LPcpp:1630-1642
    double get_curBinObj();                                   // lpbox_cur_bin_obj          LPcpp:1644-1646
    double *get_x_iters_d(int ws);                            // lpbox_get_x_iters          LPcpp:1616-1627: (n_live x ws) row-major
    int get_n();                                              // lpbox_get_n                LPh:379-381
    int get_iter();                                           // lpbox_get_iter             LPh:347-349
    double *get_x_sol();                                      // lpbox_get_x_sol            LPcpp:1648-1665: org_n entries in {0,1}
    double *get_final_x_sol();                                // lpbox_get_final_x_sol      LPcpp:1668-1685: the n_live raw values
    int check_infeasible_lpbox();                             // lpbox_check_infeasible_lpbox LPcpp:1577-1591
    int check_infeasible_l2f();                               // lpbox_check_infeasible_l2f LPcpp:1593-1612
    void set_fix_threshold(double t);                         // LPh:365-367
    void set_consistency(int c);                              // LPh:369-371
    void set_does_log(int on);                                // the per-iteration text log <root>/log/<k>_<j>_log_<i>.txt (LPh:148: ON by default in the
                                                              // reference; OFF here until asked for; lpbox_set_log / lpbox_get_log, on-chip path)

    // ---- not in the reference ----
    void set_problem(int n, int l, const int *colptr, const int *rowidx, const double *b, const double *f = nullptr);   // lpbox_set_problem_lp
    void set_data_root(const std::string &root);
    int get_org_n();
    bool on_large_path() const;
    long long outer_iterations();                             // lpbox_get_counters
    int stop_reason(int *plain_iter_plus1 = nullptr);         // lpbox_get_stop: 0 none, 1 y1_y2, 2 obj_std, 3 PCG alpha < 0, 4 all fixed

private:
    struct State;
    std::shared_ptr<State> s_;
};
