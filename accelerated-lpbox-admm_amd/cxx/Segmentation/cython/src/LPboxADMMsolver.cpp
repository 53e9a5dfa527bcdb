// LPboxADMMsolver.cpp (segmentation flavour) -- bodies of the class of LPboxADMMsolver.h over the C-ABI (include/lpbox_hip.h).
// Everything is inline: the reference's pxd includes this file textually (:1-2).
#ifndef LPBOX_SEG_SOLVER_CPP_INCLUDED
#define LPBOX_SEG_SOLVER_CPP_INCLUDED
#include "LPboxADMMsolver.h"

#include <lpbox_hip.h>

#include <sys/stat.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace {
inline bool seg_quiet() { const char *e = getenv("LPBOX_QUIET"); return e && *e && *e != '0'; }
inline void seg_throw(const char *what) { const char *m = lpbox_last_error(); throw std::runtime_error(std::string(what) + " failed: " + (m ? m : "")); }
inline int seg_ok(int rc, const char *what) { if (rc < 0) seg_throw(what); return rc; }
inline bool seg_is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
inline std::string seg_root(const char *env, const char *dflt) { const char *e = getenv(env); return (e && *e) ? e : dflt; }

// 8-bit grayscale PNG, one stored deflate block per <= 65535 bytes (no compression library needed)
inline uint32_t png_crc(const uint8_t *p, size_t n, uint32_t c = 0xFFFFFFFFu) {
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); }
    return c;
}
inline void png_chunk(FILE *f, const char *type, const std::vector<uint8_t> &data) {
    const uint32_t n = (uint32_t)data.size();
    const uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
    fwrite(len, 1, 4, f);
    fwrite(type, 1, 4, f);
    if (n) fwrite(data.data(), 1, n, f);
    uint32_t c = png_crc((const uint8_t *)type, 4);
    c = png_crc(data.data(), n, c) ^ 0xFFFFFFFFu;
    const uint8_t crc[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    fwrite(crc, 1, 4, f);
}
inline bool png_write_gray(const std::string &path, const std::vector<uint8_t> &img, int rows, int cols) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr = {(uint8_t)(cols >> 24), (uint8_t)(cols >> 16), (uint8_t)(cols >> 8), (uint8_t)cols,
                                 (uint8_t)(rows >> 24), (uint8_t)(rows >> 16), (uint8_t)(rows >> 8), (uint8_t)rows, 8, 0, 0, 0, 0};
    png_chunk(f, "IHDR", ihdr);
    std::vector<uint8_t> raw;                                   // filter byte 0 + the row
    raw.reserve((size_t)rows * (cols + 1));
    for (int r = 0; r < rows; r++) { raw.push_back(0); raw.insert(raw.end(), img.begin() + (size_t)r * cols, img.begin() + (size_t)(r + 1) * cols); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;                                      // Adler-32 of the raw stream
    for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
    for (size_t o = 0; o < raw.size() || o == 0; o += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - o);
        const bool last = o + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + o, raw.begin() + o + n);
        if (last) break;
    }
    const uint32_t ad = (b << 16) | a;
    z.push_back((uint8_t)(ad >> 24)); z.push_back((uint8_t)(ad >> 16)); z.push_back((uint8_t)(ad >> 8)); z.push_back((uint8_t)ad);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
    fclose(f);
    return true;
}
}  // namespace

struct LPboxADMMsolver::State {
    lpbox_t *h = nullptr;
    int print_info = 0, node = 10000, problem = 0;
    std::vector<double> xiters, xsol;
    ~State() { if (h) lpbox_destroy(h); }
    void create() { h = lpbox_create(LPBOX_FLAVOUR_SEG, 1, print_info); if (!h) seg_throw("lpbox_create"); }
    double scalar(const char *name) { double v = 0; seg_ok(lpbox_debug_get_scalar(h, 0, name, &v), "lpbox_debug_get_scalar"); return v; }
};

inline LPboxADMMsolver::LPboxADMMsolver() : s_(std::make_shared<State>()) { s_->create(); }
inline LPboxADMMsolver::LPboxADMMsolver(int node) : s_(std::make_shared<State>()) {
    if (!seg_quiet()) printf("Object with node is created!\n");                        // SEGcpp:639
    s_->node = node; s_->create();
}
inline LPboxADMMsolver::LPboxADMMsolver(int node, int problem) : s_(std::make_shared<State>()) {
    if (!seg_quiet()) printf("Object with node is created!\n");                        // SEGcpp:644
    s_->node = node; s_->problem = problem; s_->create();
}
inline LPboxADMMsolver::LPboxADMMsolver(int print_info, int node, int problem) : s_(std::make_shared<State>()) {
    if (!seg_quiet()) printf("Object with node is created with three inputs!\n");      // SEGcpp:651
    s_->print_info = print_info; s_->node = node; s_->problem = problem; s_->create();
}
inline std::string LPboxADMMsolver::output_path() const {
    return seg_root("LPBOX_SEG_RESULT_ROOT", "../result") + "/output_" + std::to_string(s_->problem) + ".png";       // SEGcpp:691
}

inline void LPboxADMMsolver::ADMM_bqp_unconstrained_init() {
    State &s = *s_;
    const std::string path = seg_root("LPBOX_SEG_DATA_ROOT", "../data") + "/" + std::to_string(s.problem) + ".jpg";  // SEGcpp:690
    if (!seg_quiet()) printf("The input file is: %s, and the size is: %d\n", path.c_str(), s.node);                   // SEGcpp:703
    int rows = 0, cols = 0;
    seg_ok(lpbox_read_jpeg_gray(path.c_str(), nullptr, 0, &rows, &cols), "lpbox_read_jpeg_gray");
    std::vector<unsigned char> gray((size_t)rows * cols);
    seg_ok(lpbox_read_jpeg_gray(path.c_str(), gray.data(), (long)gray.size(), &rows, &cols), "lpbox_read_jpeg_gray");
    if (!seg_quiet()) printf("Origin image size: %d X %d = %d\n", rows, cols, rows * cols);                           // SEGcpp:711
    seg_ok(lpbox_seg_set_image(s.h, gray.data(), rows, cols, s.node), "lpbox_seg_set_image");
    int sr = 0, sc = 0;
    seg_ok(lpbox_seg_get_shape(s.h, &sr, &sc), "lpbox_seg_get_shape");
    if (!seg_quiet()) printf("Reshaped image size: %d X %d = %d\n", sr, sc, sr * sc);                                 // SEGcpp:720
    seg_ok(lpbox_init(s.h), "lpbox_init");
}

inline int LPboxADMMsolver::ADMM_bqp_unconstrained_legacy() {
    State &s = *s_;
    // side-effect files (SEGcpp:1209-1216, :1270-1277, :1376), written where the reference's directories exist
    const std::string xdir = seg_root("LPBOX_SEG_XITER_ROOT", "../xiter"), rdir = seg_root("LPBOX_SEG_RESULT_ROOT", "../result");
    const bool dump = s.print_info == 1 && seg_is_dir(xdir);
    seg_ok(lpbox_set_record(s.h, dump ? 1 : 0), "lpbox_set_record");
    const auto t0 = std::chrono::steady_clock::now();
    int energy = 0;
    seg_ok(lpbox_seg_legacy(s.h, &energy), "lpbox_seg_legacy");
    const double secs = 1.0 * (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000;
    if (dump) {
        const int k = seg_ok(lpbox_seg_get_x_history(s.h, 0, 0, nullptr), "lpbox_seg_get_x_history");
        const int n = get_org_n();
        std::vector<double> X((size_t)std::max(k, 1) * n);
        if (k) seg_ok(lpbox_seg_get_x_history(s.h, 0, k, X.data()), "lpbox_seg_get_x_history");
        if (FILE *xi = fopen((xdir + "/" + std::to_string(s.problem) + ".csv").c_str(), "w+")) {
            for (int r = 0; r < k; r++) {
                fprintf(xi, "Iter%d,", r + 1);
                for (int c = 0; c < n; c++) fprintf(xi, c + 1 < n ? "%lf," : "%lf", X[(size_t)r * n + c]);
                fprintf(xi, "\n");
            }
            fclose(xi);
        }
    }
    if (seg_is_dir(rdir)) {
        int reason = 0, p1 = 0;
        seg_ok(lpbox_get_stop(s.h, 0, &reason, &p1), "lpbox_get_stop");
        const double obj = s.scalar("cur_obj"), c = s.scalar("c");
        if (FILE *al = fopen((rdir + "/xiter_all.csv").c_str(), "a")) {
            fprintf(al, "%d,%f,%f,%d,%f\n", s.problem, obj, obj + c, p1, secs);       // SEGcpp:1376
            fclose(al);
        }
    }
    return energy;
}

inline int LPboxADMMsolver::ADMM_bqp_unconstrained_l2f(int iter_start, int iter_end, double *vec, int fix_num) {
    int ret = 0;
    const int n_live = get_n();
    seg_ok(lpbox_iterate_l2f(s_->h, iter_start, iter_end, fix_num ? vec : nullptr, n_live, fix_num ? &fix_num : nullptr, &ret), "lpbox_iterate_l2f");
    return ret;
}

inline double *LPboxADMMsolver::get_x_iters_d(int ws) {
    State &s = *s_;
    const int rows = seg_ok(lpbox_get_x_iters(s.h, 0, ws, nullptr), "lpbox_get_x_iters");
    s.xiters.assign((size_t)std::max(rows, 1) * std::max(ws, 1), 0.0);
    if (rows && ws) seg_ok(lpbox_get_x_iters(s.h, 0, ws, s.xiters.data()), "lpbox_get_x_iters");
    return s.xiters.data();
}
inline int LPboxADMMsolver::get_n() { return seg_ok(lpbox_get_n(s_->h, 0), "lpbox_get_n"); }
inline int LPboxADMMsolver::get_org_n() { return seg_ok(lpbox_get_org_n(s_->h, 0), "lpbox_get_org_n"); }
inline double *LPboxADMMsolver::get_x_sol() {
    s_->xsol.assign((size_t)std::max(get_org_n(), 1), 0.0);
    seg_ok(lpbox_get_x_sol(s_->h, 0, s_->xsol.data()), "lpbox_get_x_sol");
    return s_->xsol.data();
}
inline double LPboxADMMsolver::get_final_obj() { double v = 0; seg_ok(lpbox_seg_get_obj(s_->h, &v), "lpbox_seg_get_obj"); return v; }
inline long long LPboxADMMsolver::outer_iterations() { long long o = 0, p = 0; seg_ok(lpbox_get_counters(s_->h, 0, &o, &p), "lpbox_get_counters"); return o; }

inline void LPboxADMMsolver::save_img() {                      // white where x >= 0.5; x is the scaled image column by column (SEGcpp:822-826)
    int rows = 0, cols = 0;
    seg_ok(lpbox_seg_get_shape(s_->h, &rows, &cols), "lpbox_seg_get_shape");
    const double *x = get_x_sol();
    std::vector<uint8_t> img((size_t)rows * cols);
    for (int c = 0; c < cols; c++) for (int r = 0; r < rows; r++) img[(size_t)r * cols + c] = x[(size_t)c * rows + r] >= 0.5 ? 255 : 0;
    const std::string out = output_path();
    if (!png_write_gray(out, img, rows, cols)) throw std::runtime_error("cannot write " + out);
    if (!seg_quiet()) printf("sucessful write image: %s\n", out.c_str());            // SEGcpp:836
}
#endif  // LPBOX_SEG_SOLVER_CPP_INCLUDED
