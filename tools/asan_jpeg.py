"""Build tools/asan_jpeg.cpp with AddressSanitizer + UBSan and run the JPEG reader over damaged files: truncations, flipped bytes, and
every sampling-factor combination patched into the frame header of a colour and a grayscale file (the case the round-2 fuzzing missed)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accelerated-lpbox-admm_amd"), os.path.join(ROOT, "tests")]
from test_capi_and_host import _patch_sof_sampling
exe = os.path.join(tempfile.gettempdir(), "asan_jpeg")
subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17",
                       os.path.join(ROOT, "tools", "asan_jpeg.cpp"), os.path.join(ROOT, "accelerated-lpbox-admm_amd", "csrc", "lpbox_jpeg_host.cpp"),
                       "-o", exe])
src = open(os.path.join(ROOT, "tests", "golden", "seg", "0.jpg"), "rb").read()
rng = np.random.RandomState(7)
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
with tempfile.TemporaryDirectory() as td:
    files = []
    def put(data):
        p = os.path.join(td, "f%05d.jpg" % len(files)); open(p, "wb").write(data); files.append(p)
    for c0 in range(0x11, 0x45):                       # all (h, v) in 1..4 for component 0 x a few for component 1
        if (c0 >> 4) in (1, 2, 3, 4) and (c0 & 15) in (1, 2, 3, 4):
            for c1 in (0x11, 0x22, 0x41, 0x14, 0x44):
                put(_patch_sof_sampling(_patch_sof_sampling(src, 0, c0), 1, c1))
                put(_patch_sof_sampling(_patch_sof_sampling(src, 0, c0), 1, c1)[: len(src) // 3] + b"\xff\xd9")
    try:
        from PIL import Image
        g = os.path.join(td, "g.jpg"); Image.fromarray((rng.rand(45, 70) * 255).astype(np.uint8)).save(g, quality=90)
        gs = open(g, "rb").read()
        for hv in range(0x11, 0x45):
            if (hv >> 4) in (1, 2, 3, 4) and (hv & 15) in (1, 2, 3, 4):
                put(_patch_sof_sampling(gs, 0, hv))
    except ImportError:
        pass
    for t in range(n_trials):
        m = bytearray(src)
        if t % 3 == 0:
            m = m[: rng.randint(0, len(m))]
        else:
            for _ in range(rng.randint(1, 8)):
                m[rng.randint(0, 700 if t % 3 == 1 else len(m))] = rng.randint(0, 256)
        put(bytes(m))
    for k in range(0, len(files), 500):
        subprocess.check_call([exe] + files[k:k + 500])
print("clean:", len(files), "files")
