"""Binding of the native whole-proof entry points (include/bzh2.h: bzh_pk_create / bzh_prove_batch).

The reference-side call this stands behind is halo2_proofs::plonk::{keygen_pk, create_proof} (benches/shot.rs:58-71,
benches/board.rs:51-71); the circuit crosses the boundary as data (bzh2.circuit_data.serialize_circuit, or a blob of bzh2.circuits.CircuitLayout)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import CURVE_SCALAR_FIELD, FORM_CANONICAL, FORM_MONTGOMERY, MEM_DEVICE, MEM_HOST, Bases, BzhError, Context, int_to_limbs, load
from .circuit_data import MODULI, Circuit, serialize_circuit

_VP = ctypes.c_void_p


def _bind():
    L = load()
    if getattr(L, "_bzh_native_bound", False):
        return L
    L.bzh_pk_create.argtypes = [_VP, _VP, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_VP)]
    L.bzh_pk_free.argtypes = [_VP, _VP]
    L.bzh_pk_set_lagrange.argtypes = [_VP, _VP]
    L.bzh_pk_info.argtypes = [_VP, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32),
                              ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.bzh_verify_batch.argtypes = [_VP, _VP, ctypes.c_size_t, _VP, ctypes.c_size_t, _VP, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), _VP,
                                   ctypes.POINTER(ctypes.c_int)]
    L.bzh_prove_batch.argtypes = [_VP, _VP, ctypes.c_size_t, _VP, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_char_p,
                                  ctypes.c_size_t, _VP, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.bzh_prove_batch_seeded.argtypes = [_VP, _VP, ctypes.c_size_t, _VP, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_char_p,
                                         _VP, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.bzh_rng_expand.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_size_t, _VP]
    L.bzh_pk_vk_repr.argtypes = [_VP, _VP, ctypes.POINTER(ctypes.c_int)]
    L._bzh_native_bound = True
    return L


def rng_expand(seed: bytes, first_draw: int, draws: int) -> bytes:
    """draws [first_draw, first_draw + draws) of the ChaCha20 stream bzh_prove_batch_seeded derives from `seed` (64 bytes each)"""
    assert len(seed) == 32
    out = np.zeros(draws * 64, dtype=np.uint8)
    rc = _bind().bzh_rng_expand(seed, first_draw, draws, _VP(out.ctypes.data))
    if rc:
        raise BzhError(rc, "bzh_rng_expand")
    return out.tobytes()


QUOTIENT_INTERPRETER, QUOTIENT_BUILTIN, QUOTIENT_MODULE = 0, 1, 2

_QUOTIENT_CODE = {}   # source hash -> code object (several keys of one circuit in a process share one compilation)


def compile_quotient_source(src: str, cache_dir: str | None = None, use_cache: bool = True) -> bytes | None:
    """hipcc --genco of a bzh_pk_quotient_source text against csrc/field.cuh; the code object, or None."""
    import hashlib
    import os
    import shutil
    import subprocess
    import tempfile
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256(src.encode())
    for hdr in ("field.cuh", "field_mul_fips.inc"):   # the code object also depends on the arithmetic it is compiled against
        try:
            h.update(open(os.path.join(pkg, "csrc", hdr), "rb").read())
        except OSError:
            return None
    key = h.hexdigest()[:24]
    if use_cache and key in _QUOTIENT_CODE:
        return _QUOTIENT_CODE[key]
    cache_dir = cache_dir or os.environ.get("BZH_CACHE_DIR") or os.path.join(os.path.dirname(pkg), ".bzh2_cache")
    path = os.path.join(cache_dir, "quotient_%s_gfx950.hsaco" % key)
    if use_cache and os.path.exists(path):
        _QUOTIENT_CODE[key] = open(path, "rb").read()
        return _QUOTIENT_CODE[key]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return None
    with tempfile.TemporaryDirectory() as td:
        srcp, outp = os.path.join(td, "quotient.hip"), os.path.join(td, "quotient.hsaco")
        open(srcp, "w").write(src)
        # hipcc is a driver that execs clang through a shell: it must not inherit a profiler's preload (rocprofv3 sets LD_PRELOAD /
        # tool-library variables; a preloaded tool initialises the GPU in every child and the exec chain is then refused on this pool)
        env = {k: v for k, v in os.environ.items()
               if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") and not k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_"))}
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--genco", "-I", os.path.join(pkg, "csrc"), srcp, "-o", outp],
                           capture_output=True, text=True, env=env)
        if r.returncode != 0 or not os.path.exists(outp):
            return None
        code = open(outp, "rb").read()
    try:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = "%s.tmp%d" % (path, os.getpid())
        open(tmp, "wb").write(code)
        os.replace(tmp, path)
    except OSError:
        pass
    _QUOTIENT_CODE[key] = code
    return code


class NativeProvingKey:
    """keygen_pk on the device.  g: the n SRS points, w / u: Params.w / Params.u (affine canonical int pairs)."""

    def __init__(self, ctx: Context, circuit: Circuit, curve: int, g=None, w=None, u=None, vk_repr: int = 0x1234, window_bits: int = 0,
                 params=None):
        """SRS either as explicit points (g, w, u) or as a bzh2.params.Params (Params::new(k): its tables are used as they
        are, and commitments the upstream prover makes in the Lagrange basis are made in it)."""
        self.ctx, self.c, self.curve = ctx, circuit, curve
        self.field = CURVE_SCALAR_FIELD[curve]
        self.p = MODULI[self.field]
        self._own_bases = params is None
        if params is None:
            tbl = np.stack([np.concatenate([int_to_limbs(pt[0]), int_to_limbs(pt[1])]) for pt in list(g) + [u, w]])
            self.bases: Bases = ctx.upload_bases(curve, tbl).precompute(window_bits)
            self._g0_u_w = np.ascontiguousarray(np.stack([tbl[0], tbl[-2], tbl[-1]]))
        else:
            import weakref
            self._params = params                      # keep the borrowed tables alive as long as the key
            params._borrowers.append(weakref.ref(self))
            self.bases = params.bases
            gp, _, wp, up, _ = params.points(want_lagrange=False)
            self._g0_u_w = np.ascontiguousarray(np.stack([gp[0], np.concatenate([int_to_limbs(up[0]), int_to_limbs(up[1])]),
                                                          np.concatenate([int_to_limbs(wp[0]), int_to_limbs(wp[1])])]))
        # a circuit built by the C++ front end (bzh2.circuits.CircuitLayout.blob()) arrives already serialised
        blob = bytes(circuit) if isinstance(circuit, (bytes, bytearray)) else serialize_circuit(circuit, self.p, vk_repr)
        L = _bind()
        h = _VP()
        ctx._check(L.bzh_pk_create(ctx.handle, self.bases.handle, blob, len(blob), ctypes.byref(h)), "bzh_pk_create")
        self.handle = h
        rb, mp = ctypes.c_size_t(), ctypes.c_size_t()
        na, nr, ur = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        ctx._check(L.bzh_pk_info(h, ctypes.byref(rb), ctypes.byref(mp), ctypes.byref(na), ctypes.byref(nr), ctypes.byref(ur)), "bzh_pk_info")
        self.rng_bytes, self.max_proof_bytes, self.num_advice, self.n, self.usable_rows = rb.value, mp.value, na.value, nr.value, ur.value
        if params is not None:
            self.set_lagrange(params.bases_lagrange)

    def vk_repr(self):
        """(the verifying-key digest this key absorbs first, whether it is still the library's placeholder)"""
        out, ph = (ctypes.c_uint8 * 32)(), ctypes.c_int()
        self.ctx._check(_bind().bzh_pk_vk_repr(self.handle, out, ctypes.byref(ph)), "bzh_pk_vk_repr")
        return int.from_bytes(bytes(out), "little"), bool(ph.value)

    def quotient_stats(self) -> dict:
        L = _bind()
        L.bzh_pk_quotient_stats.argtypes = [_VP] + [ctypes.POINTER(ctypes.c_uint32)] * 4
        v = [ctypes.c_uint32() for _ in range(4)]
        self.ctx._check(L.bzh_pk_quotient_stats(self.handle, *[ctypes.byref(x) for x in v]), "bzh_pk_quotient_stats")
        return {"ops": v[0].value, "multiplications_per_row": v[1].value, "lds_slots": v[2].value, "hoisted_columns": v[3].value}

    # ---- the quotient evaluator as compiled code -------------------------------------------------------------
    def quotient_source(self) -> str:
        """the evaluator program as straight-line HIP source (BzhError E_RANGE if the circuit does not fit the evaluator)"""
        L = _bind()
        L.bzh_pk_quotient_source.argtypes = [_VP, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        n = ctypes.c_size_t()
        self.ctx._check(L.bzh_pk_quotient_source(self.handle, None, 0, ctypes.byref(n)), "bzh_pk_quotient_source")
        buf = ctypes.create_string_buffer(n.value + 1)
        self.ctx._check(L.bzh_pk_quotient_source(self.handle, buf, n.value + 1, ctypes.byref(n)), "bzh_pk_quotient_source")
        return buf.value.decode()

    def set_quotient_module(self, code_object: bytes | None):
        L = _bind()
        L.bzh_pk_set_quotient_module.argtypes = [_VP, _VP, ctypes.c_char_p, ctypes.c_size_t]
        self.ctx._check(L.bzh_pk_set_quotient_module(self.ctx.handle, self.handle, code_object, len(code_object) if code_object else 0),
                        "bzh_pk_set_quotient_module")

    def quotient_selected(self):
        """(flavour, builtin_available): which quotient evaluator the key runs -- QUOTIENT_INTERPRETER (k_expr_vm2),
        QUOTIENT_BUILTIN (kernel generated when libbzh2.so was built: the reference's Shot / Board circuits) or
        QUOTIENT_MODULE (a code object installed with set_quotient_module)"""
        L = _bind()
        L.bzh_pk_quotient_selected.argtypes = [_VP, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        f, b = ctypes.c_int(), ctypes.c_int()
        self.ctx._check(L.bzh_pk_quotient_selected(self.handle, ctypes.byref(f), ctypes.byref(b)), "bzh_pk_quotient_selected")
        return f.value, bool(b.value)

    def quotient_select(self, flavour: int):
        L = _bind()
        L.bzh_pk_quotient_select.argtypes = [_VP, ctypes.c_int]
        self.ctx._check(L.bzh_pk_quotient_select(self.handle, flavour), "bzh_pk_quotient_select")

    def compile_quotient(self, cache_dir: str | None = None) -> bool:
        """Make the key run its quotient program as compiled code.  The reference's circuits have a kernel inside libbzh2.so
        (generated at build time): selected, nothing to compile.  Any other circuit: quotient_source() through hipcc (a child
        process; ~5 s, cached on disk by the hash of the source), installed with set_quotient_module.  False (the interpreter
        stays) when there is neither a builtin kernel nor a working hipcc."""
        if self.quotient_selected()[1]:
            self.quotient_select(QUOTIENT_BUILTIN)
            return True
        try:
            src = self.quotient_source()
        except BzhError:
            return False          # the circuit does not fit the VM v2 evaluator
        for attempt in (0, 1):
            code = compile_quotient_source(src, cache_dir, use_cache=attempt == 0)
            if code is None:
                return False
            try:
                self.set_quotient_module(code)
                return True
            except BzhError:
                continue   # a stale or damaged cached code object: compile afresh once
        return False

    def set_lagrange(self, bases_lagrange):
        """Params::commit_lagrange for the columns upstream commits in the Lagrange basis (None: back to coefficients)."""
        self.ctx._check(_bind().bzh_pk_set_lagrange(self.handle, bases_lagrange.handle if bases_lagrange is not None else None),
                        "bzh_pk_set_lagrange")

    def close(self):
        """bzh_pk_free.  Refused (BzhError, the key stays open) while a prove / verify call on the key is still running on
        another ctx: the key's workspaces and code belong to every ctx that uses it."""
        if self.handle is not None:
            self.ctx._check(_bind().bzh_pk_free(self.ctx.handle, self.handle), "bzh_pk_free")
            self.handle = None
            if self._own_bases:
                self.bases.free()

    def _instances(self, instances):
        B = len(instances)
        rows = max([len(col) for cols in instances for col in cols] + [0])
        inst = np.zeros((B, max(len(instances[0]), 1), max(rows, 1), 4), dtype=np.uint64)
        for b, cols in enumerate(instances):
            for i, col in enumerate(cols):
                for r, v in enumerate(col):
                    inst[b, i, r] = int_to_limbs(int(v) % self.p)
        return inst, rows

    def verify_batch(self, instances, proofs, ctx: Context | None = None) -> list:
        """plonk::verify_proof for each (instances[b], proofs[b]); returns a list of bools.  ctx: the context (stream) to run on --
        the key is shared, read-only state; every ctx that uses it gets its own workspace inside the library."""
        L = _bind()
        ctx = ctx or self.ctx
        B = len(proofs)
        inst, rows = self._instances(instances)
        stride = max(max(len(pr) for pr in proofs), 1)
        buf = np.zeros((B, stride), dtype=np.uint8)
        lens = (ctypes.c_size_t * B)(*[len(pr) for pr in proofs])
        for b, pr in enumerate(proofs):
            buf[b, :len(pr)] = np.frombuffer(pr, dtype=np.uint8)
        res = (ctypes.c_int * B)()
        rc = L.bzh_verify_batch(ctx.handle, self.handle, B, _VP(inst.ctypes.data), rows, _VP(buf.ctypes.data), stride, lens,
                                _VP(self._g0_u_w.ctypes.data), res)
        ctx._check(rc, "bzh_verify_batch")
        return [bool(v) for v in res]

    def prove_batch(self, advice, instances, rng_list, device_ptr: int | None = None, seeds=None, ctx: Context | None = None) -> list:
        """advice: (B, num_advice, n, 4) uint64 canonical host array (or, with device_ptr, a device pointer to Montgomery
        limbs of that shape); instances: B lists of instance columns (equal lengths); rng_list: B byte strings of
        rng_bytes each -- or None with seeds = B 32-byte strings: the library expands each into its proof's stream on the
        device (bzh_prove_batch_seeded; the same proofs as rng_list = [rng_expand(s, 0, rng_bytes // 64) for s in seeds]).
        ctx: the context (device stream) to prove on, default the one the key was created through; several host threads may
        prove on ONE key concurrently, each through its own ctx."""
        L = _bind()
        ctx = ctx or self.ctx
        seeded = rng_list is None
        if seeded:
            assert seeds is not None and all(len(sd) == 32 for sd in seeds)
            rng_list = list(seeds)
        B = len(rng_list)
        stride = len(rng_list[0])
        assert all(len(r) == stride for r in rng_list) and (seeded or stride >= self.rng_bytes), (stride, self.rng_bytes)
        inst, rows = self._instances(instances)
        proofs = np.zeros((B, self.max_proof_bytes), dtype=np.uint8)
        lens = (ctypes.c_size_t * B)()
        if device_ptr is not None:
            adv_p, form, mem = _VP(device_ptr), FORM_MONTGOMERY, MEM_DEVICE
        else:
            a = np.ascontiguousarray(advice, dtype=np.uint64)
            assert a.shape == (B, self.num_advice, self.n, 4), a.shape
            adv_p, form, mem = _VP(a.ctypes.data), FORM_CANONICAL, MEM_HOST
        if seeded:
            rc = L.bzh_prove_batch_seeded(ctx.handle, self.handle, B, adv_p, form, mem, _VP(inst.ctypes.data), rows,
                                          b"".join(rng_list), _VP(proofs.ctypes.data), self.max_proof_bytes, lens)
        else:
            rc = L.bzh_prove_batch(ctx.handle, self.handle, B, adv_p, form, mem, _VP(inst.ctypes.data), rows, b"".join(rng_list), stride,
                                   _VP(proofs.ctypes.data), self.max_proof_bytes, lens)
        ctx._check(rc, "bzh_prove_batch")
        return [bytes(proofs[b, :lens[b]]) for b in range(B)]
