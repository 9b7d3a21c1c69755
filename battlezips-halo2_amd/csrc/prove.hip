// Whole-proof entry points: bzh_pk_create / bzh_prove_batch (boundary row b:
// "a coarser seam (whole create_proof) is what batching needs").
//
// Native counterpart of halo2_proofs 0.2.0 `plonk::{keygen_pk, create_proof}` (UPSTREAM, un-vendored:
// Cargo.lock:382-385) as called by the reference at benches/shot.rs:58-71, benches/board.rs:51-71,
// src/circuits/shot.rs:915-930, src/circuits/board.rs:907-922: one call proves `batch` independent
// witnesses of one circuit in lockstep -- every MSM, NTT, gate evaluation, scan and IPA round is ONE launch
// carrying all of them -- with the protocol, message order and randomness draw order per proof of
// create_proof, so each proof is byte-identical to proving its witness alone (the staged test drivers tests/helpers/prover_dev.py,
// oracle/halo2_oracle.py) under the same randomness stream.
//
// The circuit arrives as DATA (serialised constraint system + fixed assignment, format below): the reference's
// own constraint systems include 19 gates of the halo2_gadgets crate that is not on disk.  Host work left:
// transcripts (Blake2b), the lookup sort, challenges and blinds; everything else runs on the device out of
// one grow-only arena owned by the proving key.
//
// Circuit blob, little-endian:
//   u32 magic "BZC1" or "BZC2" | u32 k | u32 num_advice | u32 num_fixed | u32 num_instance | u32 min_degree | u8[32] vk_repr
//   u32 ngates, ngates x expr                                                   (every constraint polynomial of every gate, flattened)
//   u32 nperm, nperm x (u8 kind {0 advice, 1 fixed, 2 instance}, u32 index)
//   u32 nlookups, per lookup: u32 m, m x expr (inputs), m x expr (table)
//   u32 ncopies, ncopies x (u32 col_a, u32 row_a, u32 col_b, u32 row_b)        (indices into the permutation columns)
//   num_fixed x (u32 len, len x u8[32] canonical values)                        (rows past len are zero)
//   "BZC2" only: advice, fixed, instance query lists, each u32 count, count x (u32 column, i32 rotation), in upstream's
//   query REGISTRATION order (= the order of the evaluations in the proof); written by csrc/circuits.hip
//   expr := u8 tag, then  0 const: u8[32] | 1 advice / 2 fixed / 3 instance: u32 column, i32 rotation
//                       | 4 neg: expr | 5 add: expr expr | 6 mul: expr expr | 7 scale: expr, u8[32]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "chacha.hpp"
#include "ctx.hpp"
#include "curve.cuh"
#include "fe29.cuh"

// the table of quotient kernels generated at build time (quotient_builtin.hip); absent (null) in the generator's own link
extern "C" const bzh_builtin_quotient* bzh_builtin_quotients(size_t* count) __attribute__((weak));

#include "prove_kernels.cuh"     // opens namespace bzh { namespace {
#include "quotient_program.hpp"  // the compiler (host only)
#include "proving_key.hpp"       // closes them around the global struct bzh_pk, reopens; keygen
#include "prover.hpp"
#include "verifier.hpp"

}  // namespace
}  // namespace bzh

namespace {
// degree (in units of n - 1: every column polynomial has degree < n) and tree multiplications of the quotient's terms
template <class C>
static int quotient_histogram_t(const uint8_t* circuit, size_t circuit_len, uint32_t* polys, uint32_t* muls) {
    using SF = typename bzh::CurveScalar<C>::SF;
    bzh_pk pk;
    bzh::ParsedKey<SF> po;
    const int rc = bzh::pk_parse_t<C>(circuit, circuit_len, pk, po);
    if (rc) return rc;
    bzh::Cols reg;
    bzh::quotient_registry(pk, bzh::QuotientPtrs{}, reg);
    bzh::EPool ep;
    int tinv = -1;
    const std::vector<int> terms = bzh::quotient_terms<SF>(pk, reg, ep, &tinv);
    std::vector<int> deg(ep.n.size(), 0), mu(ep.n.size(), 0);
    for (size_t i = 0; i < ep.n.size(); i++) {   // children precede parents in the pool
        const bzh::ENode& e = ep.n[i];
        switch (e.tag) {
            case bzh::EX_QUERY: deg[i] = 1; break;
            case bzh::EX_NEG: deg[i] = deg[e.a], mu[i] = mu[e.a]; break;
            case bzh::EX_SCALE: deg[i] = deg[e.a], mu[i] = mu[e.a] + 1; break;
            case bzh::EX_ADD: deg[i] = std::max(deg[e.a], deg[e.b]), mu[i] = mu[e.a] + mu[e.b]; break;
            case bzh::EX_MUL: deg[i] = deg[e.a] + deg[e.b], mu[i] = mu[e.a] + mu[e.b] + 1; break;
            default: break;
        }
    }
    for (int d = 0; d < 16; d++) polys[d] = muls[d] = 0;
    for (int t : terms) {
        const int d = std::min(deg[t], 15);
        polys[d]++;
        muls[d] += (uint32_t)mu[t];
    }
    return BZH_OK;
}
}  // namespace


extern "C" {

int bzh_pk_create(bzh_ctx* ctx, const bzh_bases* srs, const uint8_t* circuit, size_t circuit_len, bzh_pk** out) {
    if (!ctx || !srs || !circuit || !out || srs->device != ctx->device) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    switch (srs->curve) {
        case BZH_CURVE_VESTA: return bzh::pk_create_t<bzh::VestaCurve>(ctx, srs, circuit, circuit_len, out);
        case BZH_CURVE_PALLAS: return bzh::pk_create_t<bzh::PallasCurve>(ctx, srs, circuit, circuit_len, out);
    }
    return BZH_E_ARG;  // BN254 has no cube root of unity in Fr's multiplicative generator convention used here
}

namespace {
// one prove / verify call on a key: counted for the whole call, so that bzh_pk_set_quotient_module / bzh_pk_free can refuse
struct KeyCall {
    bzh_pk* pk;
    explicit KeyCall(bzh_pk* p) : pk(p) {
        std::lock_guard<std::mutex> lk(pk->mu);
        pk->calls_in_flight++;
    }
    ~KeyCall() {
        std::lock_guard<std::mutex> lk(pk->mu);
        pk->calls_in_flight--;
    }
};
}  // namespace

int bzh_pk_free(bzh_ctx* ctx, bzh_pk* pk) {
    if (!ctx || !pk) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    {
        std::lock_guard<std::mutex> lkp(pk->mu);
        if (pk->calls_in_flight) {   // another ctx is proving / verifying on this key: its arena and code would vanish under it
            ctx->last_error = "bzh_pk_free: a bzh_prove_batch / bzh_verify_batch call on this key is still running";
            return BZH_E_ARG;
        }
    }
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();   // every ctx's stream: the arenas below belong to all of them
    if (pk->dev) (void)hipFree(pk->dev);
    if (pk->hoist) (void)hipFree(pk->hoist);
    if (pk->key29) (void)hipFree(pk->key29);
    if (pk->q_module) (void)hipModuleUnload(pk->q_module);
    for (auto& kv : pk->arenas) kv.second->release();
    delete pk;
    return BZH_OK;
}

int bzh_pk_set_lagrange(bzh_pk* pk, const bzh_bases* g_lagrange) {
    if (!pk) return BZH_E_ARG;
    if (g_lagrange && ((g_lagrange->n != pk->n + 2 && g_lagrange->n != pk->n + 3) || g_lagrange->curve != pk->curve || g_lagrange->device != pk->device || !g_lagrange->pre_c))
        return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(pk->mu);
    pk->srs_lagrange = g_lagrange;
    return BZH_OK;
}

int bzh_pk_quotient_stats(bzh_pk* pk, uint32_t* ops, uint32_t* multiplications, uint32_t* lds_slots, uint32_t* hoisted_columns) {
    if (!pk) return BZH_E_ARG;
    uint32_t no = 0, nm = 0, nl = 0;
    if (pk->q_ok) {
        no = (uint32_t)pk->qprog.ops.size();
        nl = (uint32_t)pk->qprog.nlds;
        for (auto& o : pk->qprog.ops) nm += ((o.code >> 4) < 3 && ((o.code >> 2) & 3) == bzh::V2_MUL);
    }
    if (ops) *ops = no;
    if (multiplications) *multiplications = nm;
    if (lds_slots) *lds_slots = nl;
    if (hoisted_columns) *hoisted_columns = (uint32_t)pk->hoist_cols;
    return BZH_OK;
}

static int copy_text(const std::string& src, char* buf, size_t cap, size_t* len) {
    *len = src.size();
    if (buf && cap) {
        const size_t n = std::min(cap - 1, src.size());
        memcpy(buf, src.data(), n);
        buf[n] = 0;
    }
    return BZH_OK;
}

int bzh_pk_quotient_source(bzh_pk* pk, char* buf, size_t cap, size_t* len) {
    if (!pk || !len) return BZH_E_ARG;
    if (!pk->q_ok) return BZH_E_RANGE;   // the circuit does not fit VM v2
    return copy_text(bzh::program2_source(pk->qprog, pk->field), buf, cap, len);
}

int bzh_quotient_source_for_circuit(int curve, const uint8_t* circuit, size_t circuit_len, char* buf, size_t cap, size_t* len,
                                    uint64_t* program_hash) {
    if (!circuit || !len) return BZH_E_ARG;
    bzh_pk pk;
    int rc = BZH_E_ARG;
    if (curve == BZH_CURVE_VESTA) {
        bzh::ParsedKey<bzh::CurveScalar<bzh::VestaCurve>::SF> po;
        rc = bzh::pk_parse_t<bzh::VestaCurve>(circuit, circuit_len, pk, po);
    } else if (curve == BZH_CURVE_PALLAS) {
        bzh::ParsedKey<bzh::CurveScalar<bzh::PallasCurve>::SF> po;
        rc = bzh::pk_parse_t<bzh::PallasCurve>(circuit, circuit_len, pk, po);
    }
    if (rc) return rc;
    if (!pk.q_ok) return BZH_E_RANGE;
    if (program_hash) *program_hash = pk.q_hash;
    // two flavours of the same program: saturated limbs (namespace bzh_q_<hash>) and unsaturated 9 x 29-bit limbs (bzh_q29_<hash>)
    return copy_text(bzh::program2_source(pk.qprog, pk.field, true) + "\n" + bzh::program2_source29(pk.qprog, pk.field), buf, cap, len);
}

int bzh_quotient_degree_histogram(int curve, const uint8_t* circuit, size_t circuit_len, uint32_t* polys, uint32_t* muls) {
    if (!circuit || !polys || !muls) return BZH_E_ARG;
    if (curve == BZH_CURVE_VESTA) return quotient_histogram_t<bzh::VestaCurve>(circuit, circuit_len, polys, muls);
    if (curve == BZH_CURVE_PALLAS) return quotient_histogram_t<bzh::PallasCurve>(circuit, circuit_len, polys, muls);
    return BZH_E_ARG;
}

int bzh_pk_quotient_select(bzh_pk* pk, int flavour) {
    if (!pk) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(pk->mu);
    switch (flavour) {
        case BZH_QUOTIENT_INTERPRETER: break;
        case BZH_QUOTIENT_BUILTIN:
            if (!pk->q_builtin) return BZH_E_RANGE;
            break;
        case BZH_QUOTIENT_MODULE:
            if (!pk->q_fn) return BZH_E_RANGE;
            break;
        default: return BZH_E_ARG;
    }
    pk->q_select = flavour;
    return BZH_OK;
}

int bzh_pk_quotient_selected(bzh_pk* pk, int* flavour, int* builtin_available) {
    if (!pk) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(pk->mu);
    if (flavour) *flavour = pk->q_select;
    if (builtin_available) *builtin_available = pk->q_builtin != nullptr;
    return BZH_OK;
}

int bzh_pk_set_quotient_module(bzh_ctx* ctx, bzh_pk* pk, const void* code_object, size_t len) {
    if (!ctx || !pk || pk->device != ctx->device) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::lock_guard<std::mutex> lkp(pk->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (pk->calls_in_flight) {   // a call on another ctx may hold q_fn and be about to launch it
        ctx->last_error = "bzh_pk_set_quotient_module: a bzh_prove_batch / bzh_verify_batch call on this key is still running";
        return BZH_E_ARG;
    }
    if (pk->q_module) {
        (void)hipDeviceSynchronize();   // launches of the old code object queued on ANY ctx's stream
        (void)hipModuleUnload(pk->q_module);
        pk->q_module = nullptr;
        pk->q_fn = nullptr;
    }
    if (pk->q_select == BZH_QUOTIENT_MODULE) pk->q_select = pk->q_builtin ? BZH_QUOTIENT_BUILTIN : BZH_QUOTIENT_INTERPRETER;
    if (!code_object || !len) return BZH_OK;   // back to the key's default
    if (!pk->q_ok) return BZH_E_RANGE;
    hipModule_t mod = nullptr;
    BZH_HIP_TRY(ctx, hipModuleLoadData(&mod, code_object));
    hipFunction_t fn = nullptr;
    hipDeviceptr_t hsym = nullptr;
    size_t hbytes = 0;
    unsigned long long have = 0;
    const bool ok = hipModuleGetFunction(&fn, mod, "jit_quotient") == hipSuccess &&
                    hipModuleGetGlobal(&hsym, &hbytes, mod, "jit_program_hash") == hipSuccess && hbytes == 8 &&
                    hipMemcpy(&have, hsym, 8, hipMemcpyDeviceToHost) == hipSuccess && have == pk->q_hash;
    if (!ok) {
        (void)hipModuleUnload(mod);
        ctx->last_error = "bzh_pk_set_quotient_module: not a module generated from this key's quotient program";
        return BZH_E_ARG;
    }
    pk->q_module = mod;
    pk->q_fn = fn;
    pk->q_select = BZH_QUOTIENT_MODULE;
    return BZH_OK;
}

int bzh_pk_info(const bzh_pk* pk, size_t* rng_bytes_per_proof, size_t* max_proof_bytes, uint32_t* num_advice, uint32_t* n_rows,
                uint32_t* usable_rows) {
    if (!pk) return BZH_E_ARG;
    if (rng_bytes_per_proof) *rng_bytes_per_proof = pk->rng_bytes;
    if (max_proof_bytes) {
        const size_t points = (size_t)pk->na + 2 * pk->nl + pk->nsets + pk->nl + 1 + pk->npieces + 1 + 1 + 2 * (size_t)pk->k;
        const size_t scalars = pk->instance_queries.size() + pk->advice_queries.size() + pk->fixed_queries.size() + 1 +
                               pk->perm_columns.size() + 3 * (size_t)pk->nsets + 5 * (size_t)pk->nl + pk->rot_sets.size() + 2;
        *max_proof_bytes = 32 * (points + scalars);
    }
    if (num_advice) *num_advice = (uint32_t)pk->na;
    if (n_rows) *n_rows = (uint32_t)pk->n;
    if (usable_rows) *usable_rows = (uint32_t)pk->usable;
    return BZH_OK;
}

int bzh_pk_vk_repr(const bzh_pk* pk, uint8_t* out_repr32, int* is_placeholder) {
    if (!pk) return BZH_E_ARG;
    if (out_repr32) memcpy(out_repr32, pk->vk_repr, 32);
    if (is_placeholder) *is_placeholder = (pk->vk_repr[0] == BZH_VK_REPR_PLACEHOLDER && !pk->vk_repr[1] && !pk->vk_repr[2] && !pk->vk_repr[3]) ? 1 : 0;
    return BZH_OK;
}

int bzh_verify_batch(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* instances, size_t instance_rows, const uint8_t* proofs,
                     size_t proof_stride, const size_t* proof_lens, const uint64_t* g0_u_w, int* results) {
    if (!ctx || !pk || !batch || batch > 4096 || !proofs || !proof_lens || !g0_u_w || !results) return BZH_E_ARG;
    if (pk->device != ctx->device || (pk->ni && instance_rows && !instances) || instance_rows > pk->usable) return BZH_E_ARG;
    for (size_t b = 0; b < batch; b++)
        if (proof_lens[b] > proof_stride) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KeyCall in_flight(pk);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // The commitments of the key and G'_0 are computed against pk->srs: the three points the caller passes must be the
    // same SRS's, or every valid proof would be rejected without an error.  Row 0 of the window table is the raw SRS.
    std::unique_lock<std::mutex> lkp(pk->mu);
    if (pk->srs_g0_u_w.empty()) {
        uint64_t m[3 * 8];
        const size_t idx[3] = {0, pk->n, pk->n + 1};
        for (int i = 0; i < 3; i++)
            BZH_HIP_TRY(ctx, hipMemcpyAsync(&m[i * 8], pk->srs->d_xy + idx[i] * 16, 64, hipMemcpyDeviceToHost, ctx->stream));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> canon(24);
        for (int i = 0; i < 6; i++) {
            if (pk->curve == BZH_CURVE_VESTA) bzh::h_store<bzh::VestaCurve::Base>(&canon[i * 4], bzh::fe_from_mont(bzh::h_load<bzh::VestaCurve::Base>(&m[i * 4])));
            else bzh::h_store<bzh::PallasCurve::Base>(&canon[i * 4], bzh::fe_from_mont(bzh::h_load<bzh::PallasCurve::Base>(&m[i * 4])));
        }
        pk->srs_g0_u_w = std::move(canon);
    }
    if (memcmp(pk->srs_g0_u_w.data(), g0_u_w, 3 * 64) != 0) {
        ctx->last_error = "bzh_verify_batch: g0_u_w are not G_0, U, W of the SRS this key was built on";
        return BZH_E_ARG;
    }
    lkp.unlock();
    int rc = BZH_E_ARG;
    switch (pk->curve) {
        case BZH_CURVE_VESTA:
            rc = bzh::verify_batch_t<bzh::VestaCurve>(ctx, pk, batch, instances, instance_rows, proofs, proof_stride, proof_lens, g0_u_w, results);
            break;
        case BZH_CURVE_PALLAS:
            rc = bzh::verify_batch_t<bzh::PallasCurve>(ctx, pk, batch, instances, instance_rows, proofs, proof_stride, proof_lens, g0_u_w, results);
            break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

// rng_stride == 0: `rng` holds batch x 32-byte seeds (bzh_prove_batch_seeded)
static int prove_batch_entry(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                             size_t instance_rows, const uint8_t* rng, size_t rng_stride, uint8_t* proofs, size_t proof_stride,
                             size_t* proof_lens) {
    if (!ctx || !pk || !batch || batch > 4096 || !advice || !rng || !proofs || !proof_lens) return BZH_E_ARG;
    if ((form != BZH_FORM_CANONICAL && form != BZH_FORM_MONTGOMERY) || (mem != BZH_MEM_HOST && mem != BZH_MEM_DEVICE)) return BZH_E_ARG;
    if (pk->device != ctx->device || (rng_stride != 0 && rng_stride < pk->rng_bytes) || (pk->ni && instance_rows && !instances) ||
        instance_rows > pk->usable)
        return BZH_E_ARG;
    if (mem == BZH_MEM_DEVICE && form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KeyCall in_flight(pk);   // until the proofs are in the caller's buffers (prove_batch_t synchronises the stream before it returns)
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t* d_adv = (const uint32_t*)advice;
    void* staged = nullptr;
    const size_t elems = batch * (size_t)pk->na * pk->n;
    if (mem == BZH_MEM_HOST) {
        int rc = bzh::ws_ensure(ctx, 3, elems * 32 + 256, &staged);
        if (rc) return rc;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(staged, advice, elems * 32, hipMemcpyHostToDevice, ctx->stream));
        if (form == BZH_FORM_CANONICAL && (rc = bzh::field_convert(ctx, pk->field, (uint32_t*)staged, elems, 1))) return rc;
        d_adv = (const uint32_t*)staged;
    }
    switch (pk->curve) {
        case BZH_CURVE_VESTA:
            return bzh::prove_batch_t<bzh::VestaCurve>(ctx, pk, batch, d_adv, instances, instance_rows, rng, rng_stride, proofs, proof_stride,
                                                       proof_lens);
        case BZH_CURVE_PALLAS:
            return bzh::prove_batch_t<bzh::PallasCurve>(ctx, pk, batch, d_adv, instances, instance_rows, rng, rng_stride, proofs, proof_stride,
                                                        proof_lens);
    }
    return BZH_E_ARG;
}

int bzh_prove_batch(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                    size_t instance_rows, const uint8_t* rng, size_t rng_stride, uint8_t* proofs, size_t proof_stride,
                    size_t* proof_lens) {
    if (rng_stride == 0) return BZH_E_ARG;
    return prove_batch_entry(ctx, pk, batch, advice, form, mem, instances, instance_rows, rng, rng_stride, proofs, proof_stride, proof_lens);
}

int bzh_prove_batch_seeded(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                           size_t instance_rows, const uint8_t* seeds, uint8_t* proofs, size_t proof_stride, size_t* proof_lens) {
    return prove_batch_entry(ctx, pk, batch, advice, form, mem, instances, instance_rows, seeds, 0, proofs, proof_stride, proof_lens);
}

int bzh_rng_expand(const uint8_t* seed, uint64_t first_draw, size_t draws, uint8_t* out) {
    if (!seed || (!out && draws)) return BZH_E_ARG;
    uint32_t key[8], blk[16];
    memcpy(key, seed, 32);
    for (size_t i = 0; i < draws; i++) {
        bzh::chacha20_block(key, first_draw + i, blk);
        memcpy(out + i * 64, blk, 64);
    }
    return BZH_OK;
}

}  // extern "C"
