// lp_solve_main.cpp -- command-line driver of the LP solver class, the counterpart of the reference's `./test i k j`
// (LinerProgramming/LinearProgramming/cython_solver/test.cpp:10-33: instance i of the k-item / j-bid set, init, the plain loop to
// convergence, infeasible-constraint count, wall-clock).  Adds a machine-readable RESULT line for the tests.
//   usage: lp_solve <i> <k> <j> [max_iters=20000] [window=0] [print_info=0] [does_log=0]
// window > 0 runs the early-fixing entry point in windows of that many iterations without fixing anything (exercises
// ADMM_lp_iters_l2f / get_x_iters_d from C++); window < 0 runs the rule-based early fixing ADMM_lp_iters_fix over [0, max_iters).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>

#include "LPboxADMMsolver.h"

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <i> <k> <j> [max_iters] [window]\n", argv[0]); return 2; }
    const int i = atoi(argv[1]), k = atoi(argv[2]), j = atoi(argv[3]);
    const int max_iters = argc > 4 ? atoi(argv[4]) : 20000, window = argc > 5 ? atoi(argv[5]) : 0, print_info = argc > 6 ? atoi(argv[6]) : 0;
    const int does_log = argc > 7 ? atoi(argv[7]) : 0;        // the reference's per-iteration text log (LPh:148 has it ON; opt-in here)
    try {
        const auto t0 = std::chrono::steady_clock::now();
        LPboxADMMsolver solver(print_info);
        solver.set_does_log(does_log);
        solver.readFile(i, k, j);
        solver.ADMM_lp_iters_init();
        int ret = 0;
        double checksum = 0;
        if (window < 0) ret = solver.ADMM_lp_iters_fix(0, max_iters);
        else if (window == 0) ret = solver.ADMM_lp_iters(0, max_iters);
        else
            for (int a = 0; a < max_iters && !ret; a += window) {
                ret = solver.ADMM_lp_iters_l2f(a, a + window, nullptr, 0);
                const double *x = solver.get_x_iters_d(window);
                for (long e = 0; e < (long)solver.get_n() * window; e++) checksum += x[e];
            }
        const int infeasible = solver.check_infeasible_l2f();
        printf("this is feasiblibity: %d\n", infeasible);
        const double secs = 1e-3 * (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        const double *sol = solver.get_x_sol();
        long ones = 0;
        for (int v = 0; v < solver.get_org_n(); v++) ones += sol[v] != 0;
        printf("RESULT ret=%d objective=%.17g iterations=%lld stop=%d infeasible=%d ones=%ld n=%d live=%d large=%d checksum=%.17g\n", ret, -solver.cal_obj(),
               solver.outer_iterations(), solver.stop_reason(), infeasible, ones, solver.get_org_n(), solver.get_n(), solver.on_large_path() ? 1 : 0, checksum);
        printf("Time elapsed: %gs;\n", secs);
    } catch (const std::exception &e) {
        fprintf(stderr, "lp_solve: %s\n", e.what());
        return 1;
    }
    return 0;
}
