// CPU sanitizer harness for csrc/lpbox_jpeg_host.cpp (the GPU pool offers no sanitizers): decodes every file named on the command line.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 tools/asan_jpeg.cpp accelerated-lpbox-admm_amd/csrc/lpbox_jpeg_host.cpp -o /tmp/asan_jpeg
// tools/asan_jpeg.py builds it and feeds it truncated / bit-flipped / sampling-factor-patched files.
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../include/lpbox_hip.h"

static char g_err[512];
int lpbox_fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}

int main(int argc, char **argv) {
    int ok = 0, refused = 0;
    for (int a = 1; a < argc; a++) {
        int r = 0, c = 0;
        int rc = lpbox_read_jpeg_gray(argv[a], nullptr, 0, &r, &c);
        if (rc == 0 && (long)r * c > 0 && (long)r * c < 8000000) {
            std::vector<unsigned char> out((size_t)r * c);            // exact size: an overrun of the output is caught as well
            rc = lpbox_read_jpeg_gray(argv[a], out.data(), (long)out.size(), &r, &c);
        }
        rc == 0 ? ok++ : refused++;
    }
    printf("decoded %d, refused %d\n", ok, refused);
    return 0;
}
