import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from sejonggo_amd import _lib as L
lib = L.require_gpu()
st = L.stream_ptr()
if len(sys.argv) > 1:
    lib.sgo_conv_packed_variant(int(sys.argv[1]))          # schedule variant (A/B builds)
torch.manual_seed(3)
bad = 0
for (n, h, wd, with_skip) in [(1, 7, 7, True), (3, 17, 17, True), (5, 17, 17, False), (64, 7, 7, True), (7, 5, 19, True), (2, 19, 19, False),
                              (333, 17, 17, True), (9, 1, 1, True), (40, 1, 13, True), (77, 2, 2, False), (2048, 17, 17, True)]:
    x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
    w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
    b = torch.randn(256, device="cuda").half()
    skip = torch.randn(n, h, wd, 256, device="cuda").half() if with_skip else None
    sp = None if skip is None else skip.data_ptr()
    y0 = torch.full((n, h, wd, 256), 7.0, device="cuda", dtype=torch.float16)
    y1 = torch.full((n, h, wd, 256), 5.0, device="cuda", dtype=torch.float16)
    wp = torch.empty(lib.sgo_conv3x3_tower_packed_bytes(), device="cuda", dtype=torch.uint8)
    L.check(lib.sgo_conv3x3_tower_prepack_dev(w.data_ptr(), wp.data_ptr(), st))
    L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), sp, y0.data_ptr(), st))
    ok = True
    for rep in range(4):
        y1.fill_(5.0)
        L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), wp.data_ptr(), b.data_ptr(), sp, y1.data_ptr(), st))
        torch.cuda.synchronize()
        if not torch.equal(y0, y1):
            ok = False
            d = (y0.float() - y1.float()).abs()
            print("MISMATCH", (n, h, wd, with_skip), rep, float(d.max()), int((d > 0).sum()), "of", d.numel(), flush=True)
            break
    print((n, h, wd, with_skip), "ok" if ok else "BAD", flush=True)
    bad += 0 if ok else 1
print("bad", bad)
sys.exit(1 if bad else 0)
