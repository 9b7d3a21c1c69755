"""Device-time microbench of bzh_ntt at the shapes a proof issues (development tool, not part of bench.py's contract).

  python tools/ubench_ntt.py            # k = 14 proof shapes, 2^17 extended domain, 2^20, 2^22

Prints one line per shape: ms per call (HIP events on the launch stream, bzh_ctx_timings) and ns per element.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bzh2  # noqa: E402

P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001  # Pallas base field = Vesta scalar field
ZETA = pow(5, (P - 1) // 3, P)  # a primitive cube root of unity: the shape of halo2's extended-coset shift


def main():
    dev = torch.device("cuda:0")
    ctx = bzh2.Context(0)
    ctx.profile(True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    shapes = [(14, 256, "plain"), (14, 256, "inv"), (17, 64, "coset"), (17, 64, "coset_inv"), (11, 1024, "plain"),
              (20, 8, "coset"), (22, 1, "plain")]
    reps = 10
    for k, batch, kind in shapes:
        t = torch.randint(-(1 << 63), (1 << 63) - 1, (batch, 1 << k, 4), dtype=torch.int64, device=dev, generator=gen)
        t[..., 3] &= (1 << 61) - 1
        w = bzh2.field_omega(bzh2.FIELD_FP, k, bzh2.FORM_MONTGOMERY)
        shift = None
        if kind.startswith("coset"):
            R = 1 << 256
            zm = ZETA * R % P
            shift = np.array([(zm >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
        inv = kind.endswith("inv")
        for _ in range(2):
            ctx.ntt_device(bzh2.FIELD_FP, t.data_ptr(), k, batch, w, shift, inv)
        ctx.sync()
        t0 = ctx.timings()["ntt"]["ms"]
        for _ in range(reps):
            ctx.ntt_device(bzh2.FIELD_FP, t.data_ptr(), k, batch, w, shift, inv)
        ctx.sync()
        tm = ctx.timings()
        ms = (tm["ntt"]["ms"] - t0) / reps
        print("ntt k=%2d batch=%4d %-9s  %8.3f ms/call  %6.3f ns/elem" % (k, batch, kind, ms, ms * 1e6 / (batch << k)), flush=True)
        del t
    # coeff_to_extended: 2^k coefficients -> 2^(k+3) coset evaluations without the padded copy
    import ctypes
    L = bzh2.load()
    L.bzh_coeff_to_extended.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint,
                                        ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.c_int, ctypes.c_int]
    u64p = ctypes.POINTER(ctypes.c_uint64)
    for k, batch in [(14, 64), (11, 512)]:
        ek = k + 3
        src = torch.randint(0, (1 << 61) - 1, (batch, 1 << k, 4), dtype=torch.int64, device=dev, generator=gen)
        dst = torch.empty((batch, 1 << ek, 4), dtype=torch.int64, device=dev)
        w = bzh2.field_omega(bzh2.FIELD_FP, ek, bzh2.FORM_MONTGOMERY)
        zm = ZETA * (1 << 256) % P
        shift = np.array([(zm >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
        torch.cuda.synchronize()

        def call():
            rc = L.bzh_coeff_to_extended(ctx.handle, bzh2.FIELD_FP, ctypes.c_void_p(src.data_ptr()), k, ctypes.c_void_p(dst.data_ptr()), ek,
                                         batch, w.ctypes.data_as(u64p), shift.ctypes.data_as(u64p), bzh2.FORM_MONTGOMERY, bzh2.MEM_DEVICE)
            assert rc == 0
        for _ in range(2):
            call()
        ctx.sync()
        t0 = ctx.timings()["ntt"]["ms"]
        for _ in range(reps):
            call()
        ctx.sync()
        ms = (ctx.timings()["ntt"]["ms"] - t0) / reps
        print("coeff_to_extended k=%2d->%2d batch=%4d  %8.3f ms/call  %6.3f ns/out-elem" % (k, ek, batch, ms, ms * 1e6 / (batch << ek)), flush=True)


if __name__ == "__main__":
    main()
