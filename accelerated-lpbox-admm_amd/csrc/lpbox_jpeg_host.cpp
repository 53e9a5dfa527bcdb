// lpbox_jpeg_host.cpp -- grayscale read of a JPEG file without an image library (host code only, no device work).
//
// The reference loads its segmentation inputs with `cv::imread(path, 0)` (Segmentation/Segmentation/cython/src/LPboxADMMsolver.cpp:705):
// OpenCV hands the file to libjpeg with out_color_space = JCS_GRAYSCALE, i.e. the image is the LUMINANCE component of the file,
// reconstructed with libjpeg's default inverse DCT (JDCT_ISLOW, jidctint.c of the Independent JPEG Group's library) -- no colour
// conversion, no chroma upsampling.  This file restates exactly that path for baseline / extended-sequential Huffman JPEGs
// (SOF0 / SOF1, 8-bit; what the VOC2012 images and the reference's samples are): marker parsing (ITU T.81 annex B), Huffman
// decoding (annex F.2.2), dequantisation, and the ISLOW integer inverse DCT with IJG's constants and rounding, so that the result is
// bit for bit what libjpeg (and therefore PIL's draft("L") path and OpenCV's grayscale imread) produce for the Y plane.
// Pinned in tests/test_capi_and_host.py against PIL on the reference's two sample images.  Progressive / arithmetic-coded /
// 12-bit files are refused with an error, never approximated.
#include "../../include/lpbox_hip.h"
#include "lpbox_capi_internal.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Huff {                       // T.81 annex C / F.2.2.3: canonical code of one table
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    bool set = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        set = true;
    }
};

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; };

struct Reader {                     // entropy-coded segment: MSB-first bits, FF00 stuffing, stops at a marker
    const uint8_t *p, *end;
    uint32_t acc = 0; int nbits = 0;
    bool hit_marker = false;
    int bit() {
        if (nbits == 0) {
            int b = 0;
            if (p < end && !hit_marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) p++;
                    else { hit_marker = true; p--; b = 0; }      // a marker: feed zeros (libjpeg does the same on a truncated scan)
                }
            }
            acc = (uint32_t)b; nbits = 8;
        }
        nbits--;
        return (acc >> nbits) & 1;
    }
    int receive(int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bit(); return v; }
    void reset() { nbits = 0; acc = 0; hit_marker = false; }
};

inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }     // F.2.2.1

inline int decode_sym(Reader &r, const Huff &h) {               // F.2.2.3 DECODE
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | r.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

const int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// IJG jidctint.c, jpeg_idct_islow: CONST_BITS = 13, PASS1_BITS = 2, constants FIX(x) = round(x * 2^13)
constexpr int CB = 13, P1 = 2;
constexpr int32_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                  F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
// (64-bit temporaries: identical values for every legal coefficient block -- IJG's INT32 arithmetic does not overflow on those -- and no
//  signed overflow on a corrupt file, whose coefficients are clamped to +-2^20 before they get here)
inline int64_t descale(int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; }

inline void idct_1d(const int64_t in[8], int64_t out[8], int shift) {
    int64_t z2 = in[2], z3 = in[6];
    int64_t z1 = (z2 + z3) * F0541;
    int64_t tmp2 = z1 + z3 * (-F1847);
    int64_t tmp3 = z1 + z2 * F0765;
    z2 = in[0]; z3 = in[4];
    int64_t tmp0 = (z2 + z3) * (1 << CB);
    int64_t tmp1 = (z2 - z3) * (1 << CB);
    const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3;
    const int64_t z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}

inline void idct_islow(const int32_t coef[64], uint8_t *dst, long stride) {
    int64_t ws[64];
    for (int c = 0; c < 8; c++) {                                   // pass 1: columns
        int64_t in[8], out[8];
        bool ac = false;
        for (int r = 0; r < 8; r++) { in[r] = coef[r * 8 + c]; if (r && in[r]) ac = true; }
        if (!ac) { const int64_t dc = in[0] * (1 << P1); for (int r = 0; r < 8; r++) ws[r * 8 + c] = dc; continue; }
        idct_1d(in, out, CB - P1);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = out[r];
    }
    for (int r = 0; r < 8; r++) {                                   // pass 2: rows, descale by 2^(CONST_BITS + PASS1_BITS + 3), centre, clamp
        int64_t out[8];
        idct_1d(ws + r * 8, out, CB + P1 + 3);
        for (int c = 0; c < 8; c++) { const int64_t v = out[c] + 128; dst[r * stride + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    }
}

inline int32_t clamp20(int64_t v) { return (int32_t)(v < -(1 << 20) ? -(1 << 20) : (v > (1 << 20) ? (1 << 20) : v)); }   // never binds on a legal stream

int fail_jpeg(const char *path, const char *what) { return lpbox_fail(LPBOX_E_BADARG, "%s: %s", path, what); }

}  // namespace

extern "C" int lpbox_read_jpeg_gray(const char *path, unsigned char *out, long cap, int *rows, int *cols) {
    if (!path) return lpbox_fail(LPBOX_E_BADARG, "null path");
    FILE *fp = fopen(path, "rb");
    if (!fp) return lpbox_fail(LPBOX_E_IO, "cannot open %s", path);
    std::vector<uint8_t> buf;
    {
        uint8_t tmp[65536]; size_t k;
        while ((k = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + k);
        fclose(fp);
    }
    const size_t n = buf.size();
    if (n < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return fail_jpeg(path, "not a JPEG file (no SOI)");
    uint16_t Q[4][64] = {{0}};
    bool qset[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    int H = 0, W = 0, restart = 0;
    size_t i = 2;
    bool have_sof = false;
    while (i + 4 <= n) {
        if (buf[i] != 0xFF) return fail_jpeg(path, "marker expected");
        while (i < n && buf[i] == 0xFF) i++;                        // fill bytes
        if (i >= n) break;
        const int m = buf[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (i + 2 > n) return fail_jpeg(path, "truncated segment");
        const size_t L = ((size_t)buf[i] << 8) | buf[i + 1];
        if (L < 2 || i + L > n) return fail_jpeg(path, "bad segment length");
        const uint8_t *s = &buf[i + 2]; const size_t sl = L - 2;
        if (m == 0xDB) {                                            // DQT
            size_t k = 0;
            while (k < sl) {
                const int pq = s[k] >> 4, tq = s[k] & 15; k++;
                if (tq > 3 || k + (pq ? 128 : 64) > sl) return fail_jpeg(path, "bad DQT");
                for (int e = 0; e < 64; e++) { Q[tq][e] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
                qset[tq] = true;
            }
        } else if (m == 0xC4) {                                     // DHT
            size_t k = 0;
            while (k < sl) {
                if (k + 17 > sl) return fail_jpeg(path, "bad DHT");
                const int tc = s[k] >> 4, th = s[k] & 15; k++;
                if (tc > 1 || th > 3) return fail_jpeg(path, "bad DHT class / id");
                Huff &h = tc ? ac[th] : dc[th];
                int cnt = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[k++]; cnt += h.bits[l]; }
                if (cnt > 256 || k + cnt > sl) return fail_jpeg(path, "bad DHT counts");
                memcpy(h.vals, s + k, cnt); k += cnt;
                h.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {                        // SOF0 / SOF1: sequential Huffman
            if (sl < 6 || s[0] != 8) return fail_jpeg(path, "only 8-bit sequential JPEG is supported");
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
            const int nc = s[5];
            if (H <= 0 || W <= 0 || nc < 1 || nc > 4 || sl < 6 + 3 * (size_t)nc) return fail_jpeg(path, "bad SOF");
            if ((long)H * W > (1L << 26)) return fail_jpeg(path, "image larger than 64 Mpixel");
            comps.resize(nc);
            for (int c = 0; c < nc; c++) {
                comps[c].id = s[6 + 3 * c]; comps[c].h = s[7 + 3 * c] >> 4; comps[c].v = s[7 + 3 * c] & 15; comps[c].tq = s[8 + 3 * c] & 3;
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4) return fail_jpeg(path, "bad sampling factors");
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return fail_jpeg(path, "progressive / lossless / arithmetic-coded JPEG is not supported");
        } else if (m == 0xDD) {                                     // DRI
            if (sl < 2) return fail_jpeg(path, "bad DRI");
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {                                     // SOS: the one scan of a sequential file
            if (!have_sof) return fail_jpeg(path, "SOS before SOF");
            const int ns = s[0];
            if (ns != (int)comps.size() || sl < 1 + 2 * (size_t)ns + 3) return fail_jpeg(path, "non-interleaved scans are not supported");
            for (int c = 0; c < ns; c++) {
                const int id = s[1 + 2 * c];
                if (id != comps[c].id) return fail_jpeg(path, "scan component order differs from the frame's");
                comps[c].td = s[2 + 2 * c] >> 4; comps[c].ta = s[2 + 2 * c] & 15;
                if (comps[c].td > 3 || comps[c].ta > 3 || !dc[comps[c].td].set || !ac[comps[c].ta].set || !qset[comps[c].tq])
                    return fail_jpeg(path, "scan refers to a table that was not defined");
            }
            if (rows) *rows = H;
            if (cols) *cols = W;
            if (!out) return LPBOX_OK;                              // size query
            if (cap < (long)H * W) return lpbox_fail(LPBOX_E_BADARG, "buffer too small for a %d x %d image", H, W);
            // T.81 A.2.2 / libjpeg (comps_in_scan == 1): the MCU of a one-component scan is ONE 8x8 block in raster order,
            // whatever sampling factors the frame header states
            if (comps.size() == 1) comps[0].h = comps[0].v = 1;
            int hmax = 1, vmax = 1;
            for (auto &c : comps) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
            // libjpeg would upsample a luminance plane that is not the most finely sampled component; no encoder in use writes
            // such files, so they are refused rather than approximated (the plane below is sized from component 0)
            if (comps[0].h != hmax || comps[0].v != vmax) return fail_jpeg(path, "luminance is subsampled: not supported");
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            const Comp &Y = comps[0];
            const long pw = (long)mcux * Y.h * 8, ph = (long)mcuy * Y.v * 8;       // padded luminance plane
            std::vector<uint8_t> plane((size_t)pw * ph);
            Reader r{&buf[i + L], buf.data() + n};
            int mcu_left = restart;
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (restart && mcu_left == 0) {                 // RSTn: byte-align, skip the marker, reset the predictors
                        r.reset();
                        while (r.p + 1 < r.end && !(r.p[0] == 0xFF && r.p[1] >= 0xD0 && r.p[1] <= 0xD7)) r.p++;
                        if (r.p + 1 < r.end) r.p += 2;
                        for (auto &c : comps) c.pred = 0;
                        mcu_left = restart;
                    }
                    if (restart) mcu_left--;
                    for (size_t ci = 0; ci < comps.size(); ci++) {
                        Comp &c = comps[ci];
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                int32_t coef[64] = {0};
                                const int t = decode_sym(r, dc[c.td]);
                                if (t < 0 || t > 11) return fail_jpeg(path, "corrupt DC code");
                                c.pred = clamp20((int64_t)c.pred + extend(r.receive(t), t));
                                coef[0] = clamp20((int64_t)c.pred * Q[c.tq][0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_sym(r, ac[c.ta]);
                                    if (rs < 0) return fail_jpeg(path, "corrupt AC code");
                                    const int run = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (run == 15) { k += 16; continue; } break; }      // ZRL / EOB
                                    k += run;
                                    if (k > 63) return fail_jpeg(path, "corrupt AC run");
                                    coef[kZigzag[k]] = clamp20((int64_t)extend(r.receive(sz), sz) * Q[c.tq][k]);
                                    k++;
                                }
                                if (ci == 0) idct_islow(coef, &plane[((size_t)(my * c.v + by) * 8) * pw + (size_t)(mx * c.h + bx) * 8], pw);
                            }
                    }
                }
            if (pw < W || ph < H) return fail_jpeg(path, "internal: luminance plane smaller than the image");
            for (int y = 0; y < H; y++) memcpy(out + (size_t)y * W, &plane[(size_t)y * pw], (size_t)W);
            return LPBOX_OK;
        }
        i += L;
    }
    return fail_jpeg(path, "no scan found");
}
