// seg_solve_main.cpp -- command-line driver of the segmentation solver class, the counterpart of the reference's
// image_segmentation.cpp:24-29 (for problem in first..last: LPboxADMMsolver(print_info, numNodes, problem); init; legacy loop).
//   usage: seg_solve <numNodes> <first> <last> [print_info=0] [save=0]
// One RESULT line per problem for the tests.
#include <cstdio>
#include <cstdlib>
#include <exception>

#include "LPboxADMMsolver.h"

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <numNodes> <first> <last> [print_info] [save]\n", argv[0]); return 2; }
    const int nodes = atoi(argv[1]), first = atoi(argv[2]), last = atoi(argv[3]);
    const int print_info = argc > 4 ? atoi(argv[4]) : 0, save = argc > 5 ? atoi(argv[5]) : 0;
    try {
        for (int i = first; i <= last; i++) {
            LPboxADMMsolver solver(print_info, nodes, i);
            solver.ADMM_bqp_unconstrained_init();
            const int energy = solver.ADMM_bqp_unconstrained_legacy();
            const double *x = solver.get_x_sol();
            long ones = 0;
            for (int v = 0; v < solver.get_org_n(); v++) ones += x[v] != 0;
            if (save) solver.save_img();
            printf("RESULT problem=%d energy=%d objective=%.17g iterations=%lld n=%d ones=%ld\n", i, energy, solver.get_final_obj(),
                   solver.outer_iterations(), solver.get_org_n(), ones);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "seg_solve: %s\n", e.what());
        return 1;
    }
    return 0;
}
