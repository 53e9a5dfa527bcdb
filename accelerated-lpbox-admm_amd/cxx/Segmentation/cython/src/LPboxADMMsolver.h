// LPboxADMMsolver.h -- the reference's C++ solver class, SEGMENTATION flavour, as a thin host-side class over the C-ABI of
// liblpbox_hip.so.  Public interface of Segmentation/Segmentation/cython/src/LPboxADMMsolver.h:239-307 as bound by the pxd next
// to it (:5-18) and driven by image_segmentation.cpp:24-29: same names, arguments and return values, so that the reference's
// Cython module and its C++ driver build against THIS pair of files unchanged.  No solver arithmetic
// here: every method forwards to include/lpbox_hip.h.  Deliberate differences, as for the LP class: copies share one solver; the
// arrays behind get_x_iters_d / get_x_sol belong to the object; a failing call throws std::runtime_error; the image is read by the
// library itself (lpbox_read_jpeg_gray = cv::imread(path, 0) for sequential JPEGs), save_img writes an 8-bit grayscale PNG with
// stored (uncompressed) deflate blocks; directories: the reference's CWD-relative "../data", "../result", "../xiter"
// (SEGcpp:690-696) unless LPBOX_SEG_DATA_ROOT / LPBOX_SEG_RESULT_ROOT / LPBOX_SEG_XITER_ROOT are set; LPBOX_QUIET=1 silences stdout.
#pragma once
#include <memory>
#include <string>
#include <vector>

class LPboxADMMsolver {
public:
    LPboxADMMsolver();                                        // SEGh:241
    LPboxADMMsolver(int node);                                // SEGh:243, SEGcpp:638-641
    LPboxADMMsolver(int node, int problem);                   // SEGh:247, SEGcpp:643-648
    LPboxADMMsolver(int print_info, int node, int problem);   // SEGh:249, SEGcpp:650-655

    void ADMM_bqp_unconstrained_init();                       // lpbox_read_jpeg_gray + lpbox_seg_set_image + lpbox_init   SEGcpp:658-810
    int ADMM_bqp_unconstrained_legacy();                      // lpbox_seg_legacy           SEGcpp:1200-1380: returns int(cur_obj + c)
    int ADMM_bqp_unconstrained_l2f(int iter_start, int iter_end, double *vec, int fix_num);   // lpbox_iterate_l2f   SEGcpp:917-1195
    double *get_x_iters_d(int ws);                            // lpbox_get_x_iters          SEGh:257: (n_live x ws) row-major
    int get_n();                                              // lpbox_get_n                SEGh:259
    int get_org_n();                                          // lpbox_get_org_n            SEGh:263
    double *get_x_sol();                                      // lpbox_get_x_sol            SEGh:269: org_n entries in {0,1}
    double get_final_obj();                                   // lpbox_seg_get_obj          SEGcpp:868
    void save_img();                                          // lpbox_seg_get_shape + get_x_sol -> <result>/output_<problem>.png   SEGcpp:812-837

    // ---- not in the reference ----
    long long outer_iterations();                             // lpbox_get_counters
    std::string output_path() const;

private:
    struct State;
    std::shared_ptr<State> s_;
};
