/*
 * bzh2 -- MI355X (gfx950) prover-backend primitives for the BattleZips Halo2
 * circuits, flat C ABI.  This is the drop-in boundary: the entry points are
 * what an FFI shim inside `halo2_proofs` (the crate the reference calls into)
 * would bind for the create_proof hot path.  See INTEGRATION.md for the Rust
 * `extern "C"` block and the three call sites it replaces.
 *
 * The reference itself contains no prover arithmetic: its
 * hot path is the call
 *     create_proof(&params, &pk, &[circuit], &[&[&instances]], OsRng, &mut transcript)
 *   benches/shot.rs:68, benches/board.rs:61-68,
 *   src/circuits/shot.rs:921-928, src/circuits/board.rs:913-920,
 *   src/wasm/circuit_wasm.rs:66-73,151-158
 * which funnels into halo2_proofs 0.2.0 (Cargo.lock:382-385, un-vendored):
 *     arithmetic::best_multiexp(&[Scalar], &[Affine]) -> Curve      -> bzh_msm
 *     Params::{commit, commit_lagrange}                              -> bzh_bases_upload + bzh_msm
 *     arithmetic::best_fft(&mut [Scalar], omega, log_n)              -> bzh_ntt
 *     EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff} -> bzh_ntt (inverse / coset_shift)
 *
 * Data conventions
 *   field element : 4 x uint64 little-endian limbs (32 bytes).
 *                   form = BZH_FORM_MONTGOMERY  (x*2^256 mod p; pasta_curves' in-memory form,
 *                          so `&[Fp]` crosses the boundary without repacking) or
 *                   form = BZH_FORM_CANONICAL   (ff::PrimeField::to_repr, src/utils/binary.rs:36).
 *   affine point  : x || y (64 bytes, 8 limbs); (0,0) is the identity.
 *   Jacobian point: X || Y || Z (96 bytes, 12 limbs), x = X/Z^2, y = Y/Z^3, Z = 0 identity.
 *   mem           : BZH_MEM_HOST   - pointer is host memory, the call copies in/out and
 *                                    returns when the result is in the caller's buffer;
 *                   BZH_MEM_DEVICE - pointer is HBM on the ctx's device; the call only
 *                                    enqueues work on the ctx's stream (use bzh_ctx_sync).
 *
 * Ownership / errors / threading
 *   The caller owns every buffer; the library never keeps a host pointer past
 *   return.  Device tables are explicit handles (bzh_bases).  Every function
 *   returns 0 (BZH_OK) or a negative bzh_status and never throws or aborts
 *   across the boundary (halo2's callers `assert_eq!(coeffs.len(), bases.len())`
 *   and panic: a shim maps nonzero to panic!/Error::Synthesis).  One ctx = one
 *   device + one HIP stream; calls on one ctx are serialised by an internal
 *   mutex, several ctxs may be used from several threads.
 */
#ifndef BZH2_H
#define BZH2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bzh_ctx bzh_ctx;
typedef struct bzh_bases bzh_bases;
typedef struct bzh_transcript bzh_transcript;

typedef enum {
    BZH_OK = 0,
    BZH_E_ARG = -1,    /* null pointer, bad enum, size mismatch               */
    BZH_E_OOM = -2,    /* hipMalloc / host allocation failed                  */
    BZH_E_HIP = -3,    /* a HIP runtime call failed (see bzh_last_error)      */
    BZH_E_RANGE = -4,  /* log_n beyond the field's 2-adicity, n too large ... */
    BZH_E_NOGPU = -5,  /* no usable gfx950 device                             */
    BZH_E_VERIFY = -6  /* bzh_ipa_verify: the proof does not verify           */
} bzh_status;

typedef enum { BZH_CURVE_VESTA = 0, BZH_CURVE_PALLAS = 1, BZH_CURVE_BN254 = 2 } bzh_curve;
typedef enum { BZH_FIELD_FP = 0, BZH_FIELD_FQ = 1, BZH_FIELD_BN254_FR = 2, BZH_FIELD_BN254_FQ = 3 } bzh_field;
typedef enum { BZH_FORM_CANONICAL = 0, BZH_FORM_MONTGOMERY = 1 } bzh_form;
typedef enum { BZH_MEM_HOST = 0, BZH_MEM_DEVICE = 1 } bzh_mem;

/* kernel classes timed by bzh_ctx_profile (indices into bzh_ctx_timings) */
typedef enum {
    BZH_T_MSM_DIGITS = 0,
    BZH_T_MSM_ACCUMULATE = 1,
    BZH_T_MSM_REDUCE = 2,
    BZH_T_MSM_FINALIZE = 3,
    BZH_T_NTT = 4,
    BZH_T_POLY = 5,
    BZH_T_QUOTIENT = 6, /* the quotient's gate evaluation over the extended coset (k_expr_vm2) */
    BZH_T_COUNT = 8
} bzh_timer;

/* ---- library / context -------------------------------------------------- */
const char* bzh_version(void);
const char* bzh_strerror(int status);
/* number of HIP devices visible (0 when there is no GPU); never fails */
int bzh_device_count(void);

/* One ctx = one device + one stream owned by the ctx. */
int bzh_ctx_create(int device, bzh_ctx** out);
/* Same, but work is enqueued on a caller-owned hipStream_t (e.g. the stream a
 * host framework already uses).  The stream must outlive the ctx. */
int bzh_ctx_create_on_stream(int device, void* hip_stream, bzh_ctx** out);
int bzh_ctx_destroy(bzh_ctx* ctx);
int bzh_ctx_sync(bzh_ctx* ctx);
/* text of the last HIP error seen on this ctx ("" if none) */
const char* bzh_last_error(const bzh_ctx* ctx);

/* Per-kernel-class timing with HIP events on the ctx's stream.  enable != 0
 * starts collecting (and clears the accumulators); bzh_ctx_timings syncs the
 * stream and returns, per class, accumulated milliseconds and launch counts
 * since enabling.  ms / launches must each hold BZH_T_COUNT entries.
 * A measurement aid, off by default: the event records are stream commands too -- one proof at a time they cost ~10 % of
 * its latency (10.4 -> 11.7 ms, BoardCircuit k = 14), 0.5 % of a batch's throughput.  Leave it off in production. */
int bzh_ctx_profile(bzh_ctx* ctx, int enable);
int bzh_ctx_timings(bzh_ctx* ctx, double* ms, uint64_t* launches);
/* Algorithmic bytes (MSM 32*B*N + 64*N per launch, NTT 64*N per transform) of the launches timed since
 * profiling was enabled, per kernel class (BZH_T_COUNT entries); divides by bzh_ctx_timings' ms for GB/s. */
int bzh_ctx_work(bzh_ctx* ctx, double* algorithmic_bytes);
/* bucket additions the MSM accumulation made since profiling was enabled (one per non-zero window digit: zero scalars and
 * zero digits of small scalars cost nothing) -- the work the integer multipliers actually did, for the ALU-side roofline */
int bzh_ctx_msm_additions(bzh_ctx* ctx, uint64_t* additions);

/* ---- commitment bases (Params.g / Params.g_lagrange of halo2's IPA params) -
 * Replaces the `bases: &[C]` argument of best_multiexp for tables that live
 * across many MSMs: n affine points are copied to HBM once (converted to
 * Montgomery form if needed) and referenced by handle afterwards. */
int bzh_bases_upload(bzh_ctx* ctx, int curve, const uint64_t* xy, size_t n, int form, int mem, bzh_bases** out);
/* Expand a table in place into the fixed-base window table  row w = 2^(c*w) * G_i  (w < ceil(256/c)),
 * c = window_bits or 0 for the library's choice.  Worth it for tables that serve many MSMs (the SRS:
 * every commitment of every proof uses Params.g or Params.g_lagrange): all windows of an MSM then
 * share one bucket set and the final doubling chain disappears.  Costs ceil(256/c) x the table's HBM
 * (k=14: 25 MB) and a one-time build; results are identical with or without it. */
int bzh_bases_precompute(bzh_ctx* ctx, bzh_bases* bases, int window_bits);
/* A synthetic table made on the device: bases[i] = [i + 1] G for i < n (SURVEY 8d's "cheap generator walk" for the 2^24-point
 * MSM microbench, BASELINE.json configs[4]: 1 GB of points that never cross PCIe).  g_xy: one affine point in `form` (host).
 * The CPU oracle's orc_point_walk makes the same set. */
int bzh_bases_walk(bzh_ctx* ctx, int curve, const uint64_t* g_xy, int form, size_t n, bzh_bases** out);
/* `count` points of a table from index `first`, affine canonical x || y into out_xy (host); a window table continues through
 * its rows (row w starts at w * n).  BZH_E_RANGE past the end. */
int bzh_bases_points(bzh_ctx* ctx, const bzh_bases* bases, size_t first, size_t count, uint64_t* out_xy);
int bzh_bases_free(bzh_ctx* ctx, bzh_bases* bases);
size_t bzh_bases_len(const bzh_bases* bases);

/* ---- multi-scalar multiplication  (halo2_proofs arithmetic::best_multiexp) -
 * out[b] = sum_{i<n} scalars[b*n + i] * bases[i]   for b < batch.
 * scalars: batch*n field elements of the curve's scalar field (`form`).
 * n may be smaller than the table (prefix is used, as Params::commit does for
 * short polynomials).  out: batch Jacobian points (12 limbs each) in `form`.
 * `mem` applies to both scalars and out. */
int bzh_msm(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* scalars, size_t n, size_t batch, int form, int mem,
            uint64_t* out_xyz);

/* ---- NTT  (halo2_proofs arithmetic::best_fft + EvaluationDomain) ----------
 * In place over `batch` contiguous vectors of 2^log_n elements; natural order
 * in and out.
 *   inverse == 0 : a_i <- sum_j a_j * omega^(ij); if coset_shift != NULL element j is
 *                  first multiplied by shift^j (coeff_to_extended's zeta powers).
 *   inverse != 0 : uses omega^-1, multiplies by n^-1 (EvaluationDomain::ifft); if
 *                  coset_shift != NULL result i is multiplied by shift^-i
 *                  (extended_to_coeff).
 * omega must be a primitive 2^log_n-th root of unity; omega and coset_shift are
 * host pointers to 4 limbs in `form`. */
int bzh_ntt(bzh_ctx* ctx, int field, uint64_t* data, unsigned log_n, size_t batch, const uint64_t* omega,
            const uint64_t* coset_shift, int inverse, int form, int mem);

/* EvaluationDomain::coeff_to_extended (halo2_proofs 0.2.0 poly/domain.rs, UPSTREAM; reached from create_proof at
 * benches/shot.rs:68): `batch` polynomials of 2^log_n coefficients at `coeffs` (contiguous) -> their evaluations over
 * the 2^log_ext-point coset shift*<omega_ext> at `out` (contiguous vectors of 2^log_ext).  Same result as zero-padding
 * each polynomial to 2^log_ext and calling bzh_ntt(..., coset_shift, inverse = 0), without the padded copy: the first
 * pass reads only the 2^log_n coefficients.  `coeffs` is not written; `out` must not overlap it. */
int bzh_coeff_to_extended(bzh_ctx* ctx, int field, const uint64_t* coeffs, unsigned log_n, uint64_t* out, unsigned log_ext,
                          size_t batch, const uint64_t* omega_ext, const uint64_t* coset_shift, int form, int mem);

/* ---- prover-stage vector primitives (N4, N5, N6) ------------
 * Field elements in `form`; `mem` applies to every pointer of the call.  With BZH_MEM_DEVICE the
 * read-only inputs must already be in Montgomery form (the library does not write to them).
 *
 * bzh_batch_invert     ff::BatchInvert: data[i] <- data[i]^-1, zeros stay zero (grand-product denominators)
 * bzh_prefix_product   in place, per vector of n: out[0] = 1, out[i] = prod_{j<i} in[j]
 *                      (the running products z(X) of permutation::Argument::commit / lookup commit_product)
 * bzh_eval_polynomial  arithmetic::eval_polynomial: out[b] = sum_i coeffs[b][i] * x_b^i;
 *                      nx = 1 (one point for every polynomial) or nx = batch
 * bzh_inner_product    arithmetic::compute_inner_product per vector pair
 * bzh_fold             IPA round fold: out[b][i] = in[b][i] + u_b * in[b][half + i], i < half; nu = 1 or batch
 * bzh_vec_mul          a[i] <- a[i] * b[i]
 */
/* canonical <-> Montgomery (x * 2^256 mod p) in place; device-resident pipelines keep Montgomery form throughout */
int bzh_field_convert(bzh_ctx* ctx, int field, uint64_t* data, size_t count, int to_montgomery, int mem);
/* ff::Field::random for `count` draws from a caller-supplied RNG byte stream (host pointer, 64 bytes per draw, the
 * 8 x next_u64 of pasta_curves): out[i] = bytes[64i .. 64i+64) as a 512-bit little-endian integer mod p.
 * The prover's random polynomials (vanishing argument, IPA s(X)) are n draws each; `mem` applies to out. */
int bzh_random_field(bzh_ctx* ctx, int field, const uint8_t* rng_bytes, size_t count, int form, int mem, uint64_t* out);
int bzh_batch_invert(bzh_ctx* ctx, int field, uint64_t* data, size_t count, int form, int mem);
int bzh_prefix_product(bzh_ctx* ctx, int field, uint64_t* data, size_t n, size_t batch, int form, int mem);
int bzh_eval_polynomial(bzh_ctx* ctx, int field, const uint64_t* coeffs, size_t n, size_t batch, const uint64_t* xs, size_t nx,
                        int form, int mem, uint64_t* out);
int bzh_inner_product(bzh_ctx* ctx, int field, const uint64_t* a, const uint64_t* b, size_t n, size_t batch, int form, int mem,
                      uint64_t* out);
int bzh_fold(bzh_ctx* ctx, int field, const uint64_t* in, size_t half, size_t batch, const uint64_t* u, size_t nu, int form,
             int mem, uint64_t* out);
int bzh_vec_mul(bzh_ctx* ctx, int field, uint64_t* a, const uint64_t* b, size_t count, int form, int mem);
/* arithmetic::kate_division (multiopen): out (n-1 coefficients) = quotient of the n-coefficient p(X) by (X - x);
 * the remainder p(x) is dropped.  x: 4 limbs in `form` (host). */
int bzh_kate_division(bzh_ctx* ctx, int field, const uint64_t* coeffs, size_t n, const uint64_t* x, int form, int mem,
                      uint64_t* out);
/* `batch` divisions in one pass: vector v = coeffs + v*n elements is divided by (X - xs[v]); out holds batch x (n-1). */
int bzh_kate_division_batch(bzh_ctx* ctx, int field, const uint64_t* coeffs, size_t n, size_t batch, const uint64_t* xs, int form,
                            int mem, uint64_t* out);
/* lookup::prover::permute_expression_pair (host: one sort per lookup argument): over the first usable_rows rows,
 * out_input = the input column sorted by canonical value; out_table holds the input value wherever a new run of
 * equal inputs starts and the table's unused values elsewhere (ascending, filled from the last repeated row
 * backwards, as upstream).  BZH_E_RANGE if an input value is missing from the table. */
int bzh_permute_expression_pair(int field, const uint64_t* input, const uint64_t* table, size_t usable_rows, int form,
                                uint64_t* out_input, uint64_t* out_table);

/* ---- Fiat-Shamir transcript (host; halo2_proofs transcript::{Blake2bWrite, Challenge255}) --
 * What every create_proof caller builds first (benches/shot.rs:66-67, src/circuits/board.rs:911-912:
 * `Blake2bWrite::<_, vesta::Affine, Challenge255<_>>::init(vec![])`).  Blake2b-512, personal
 * "Halo2-Transcript"; points and scalars are passed in canonical form (8 / 4 limbs).
 *   common_*  absorb only;  write_*  absorb and append to the proof bytes (compressed point / repr);
 *   squeeze_challenge  absorbs 0x00 and returns the digest reduced as a 512-bit LE integer mod the
 *   challenge field (4 canonical limbs);  proof  borrows the bytes written so far (finalize()). */
typedef struct bzh_transcript bzh_transcript;
int bzh_transcript_new(int challenge_field, bzh_transcript** out);
int bzh_transcript_free(bzh_transcript* t);
int bzh_transcript_common_point(bzh_transcript* t, const uint64_t* xy_canonical);
int bzh_transcript_common_scalar(bzh_transcript* t, const uint64_t* s_canonical);
int bzh_transcript_write_point(bzh_transcript* t, int curve, const uint64_t* xy_canonical);
int bzh_transcript_write_scalar(bzh_transcript* t, const uint64_t* s_canonical);
int bzh_transcript_squeeze_challenge(bzh_transcript* t, uint64_t* out_canonical);
int bzh_transcript_proof(const bzh_transcript* t, const uint8_t** data, size_t* len);

/* ---- gate-expression evaluation over the extended coset (row a13: poly::Evaluator + vanishing::construct) --
 * A straight-line program evaluated at every row r < 2^log_size:
 *   operand kinds  BZH_EXPR_SLOT (idx = slot), BZH_EXPR_COLUMN (idx = column, rot = row offset, wraps mod size),
 *                  BZH_EXPR_CONST (idx = constant);
 *   ops            ADD, SUB, MUL (a, b);  NEG, COPY (a);  dst = slot < BZH_EXPR_MAX_SLOTS.
 * out[r] = slot `result_slot` after the last op.  columns: host array of ncols pointers (each 2^log_size
 * elements, `mem` says where they live); consts: nconsts elements (host).  A rotation by t rows of the
 * 2^k-row circuit is rot = t * 2^(log_size - k) on the extended domain. */
enum { BZH_EXPR_ADD = 0, BZH_EXPR_SUB = 1, BZH_EXPR_MUL = 2, BZH_EXPR_NEG = 3, BZH_EXPR_COPY = 4 };
enum { BZH_EXPR_SLOT = 0, BZH_EXPR_COLUMN = 1, BZH_EXPR_CONST = 2 };
enum { BZH_EXPR_MAX_SLOTS = 24 };
typedef struct {
    uint8_t op, dst, a_kind, b_kind;
    int32_t a_idx, b_idx, a_rot, b_rot;
} bzh_expr_op;
int bzh_expr_eval(bzh_ctx* ctx, int field, const bzh_expr_op* prog, size_t nops, const uint64_t* const* columns, size_t ncols,
                  const uint64_t* consts, size_t nconsts, unsigned log_size, int result_slot, int form, int mem, uint64_t* out);

/* The same program over `batch` vectors (independent proofs) in one launch: vector v reads column c at
 * columns[c] + v * column_strides[c] elements (0, or column_strides == NULL: the column is shared, e.g. fixed /
 * permutation columns of the proving key) and its constants at consts + v * const_stride elements (0: shared;
 * otherwise >= nconsts, e.g. per-proof challenges); out holds batch x 2^log_size results. */
int bzh_expr_eval_batch(bzh_ctx* ctx, int field, const bzh_expr_op* prog, size_t nops, const uint64_t* const* columns,
                        const size_t* column_strides, size_t ncols, const uint64_t* consts, size_t nconsts, size_t const_stride,
                        unsigned log_size, int result_slot, size_t batch, int form, int mem, uint64_t* out);

/* ---- inner-product-argument opening (halo2_proofs poly::commitment::{create_proof, verify_proof}) --
 * Step 9 of plonk::create_proof and the heart of verify_proof (benches/board.rs:80-86).
 * `bases` must hold n + 2 points: the n = 2^k SRS generators followed by U and W (Params.u, Params.w);
 * a window table (bzh_bases_precompute) is recommended: every round is two MSMs against it.
 * bzh_ipa_open   writes S, (L_j, R_j) for j < k, then the scalars c and f to the transcript, exactly
 *                the upstream message order.  poly: n coefficients (`form`, `mem`); blind, x3: 4 canonical
 *                limbs (host); rng: 64 bytes of RNG output per drawn scalar in upstream's draw order
 *                (n for s(X), 1 for its blind, 2 per round) = 64 * (n + 1 + 2k) bytes; out_v = p(x3).
 * bzh_ipa_verify checks  sum_j (u_j^-1 L_j + u_j R_j) + P - [v]G_0 + [xi]S == [c]G'_0 + [c b_0 z]U + [f]W
 *                against a transcript in the same state the prover's was before bzh_ipa_open;
 *                g0_u_w: G_0, U, W as 3 canonical affine points.  Returns BZH_OK or BZH_E_VERIFY. */
int bzh_ipa_open(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* poly, int form, int mem, const uint64_t* blind,
                 const uint64_t* x3, const uint8_t* rng, size_t rng_len, bzh_transcript* transcript, uint64_t* out_v);
/* `batch` independent openings against the same bases, advanced in lockstep (every round is one MSM launch of
 * 2*batch vectors and one host round trip): polys = batch x n coefficients, blinds / x3s / out_v = batch x 4 limbs,
 * proof b draws from rng + b*rng_stride and writes to transcripts[b].  bzh_ipa_open is the batch = 1 case. */
int bzh_ipa_open_batch(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* polys, int form, int mem, size_t batch,
                       const uint64_t* blinds, const uint64_t* x3s, const uint8_t* rng, size_t rng_stride,
                       bzh_transcript* const* transcripts, uint64_t* out_v);
int bzh_ipa_verify(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* commitment_xy, const uint64_t* x3, const uint64_t* v,
                   const uint8_t* proof, size_t proof_len, bzh_transcript* transcript, const uint64_t* g0_u_w);

/* ---- whole proofs (halo2_proofs plonk::{keygen_pk, create_proof}; reference call sites benches/shot.rs:58-71,
 * benches/board.rs:51-71, src/circuits/shot.rs:915-930, src/circuits/board.rs:907-922) ----------------------
 * bzh_pk_create    keygen_pk for a circuit given as data (serialised constraint system + fixed assignment; the
 *                  format is documented at the top of csrc/prove.hip; produced by bzh_circuit_blob for the reference's circuits and by
 *                  bzh2.circuit_data.serialize_circuit for test circuits):
 *                  fixed / permutation polynomials in Lagrange, coefficient and extended-coset form, l_0 / l_last /
 *                  l_blind, resident on the device.  `srs`: n + 2 points G_0..G_(n-1), U, W with a window table
 *                  (bzh_bases_precompute).  LIFETIME: the key BORROWS `srs` (and the table given to
 *                  bzh_pk_set_lagrange): both must outlive the key -- free the key before bzh_params_free / bzh_bases_free.
 * bzh_prove_batch  create_proof for `batch` independent witnesses of that circuit in lockstep (one launch per kernel
 *                  class per protocol phase for all of them); every proof is byte-identical to proving its witness
 *                  alone.  advice: batch x num_advice x n field elements (`form`, `mem`; rows past the usable ones are
 *                  overwritten with blinding values); instances: batch x num_instance x instance_rows canonical
 *                  elements (host); rng: per proof, at offset b * rng_stride, 64 bytes per Field::random draw in
 *                  upstream's draw order (bzh_pk_info reports the byte count); proofs: batch records of
 *                  proof_stride bytes, lengths in proof_lens.  BZH_E_RANGE: a witness does not satisfy the circuit
 *                  (surplus quotient coefficients) or a lookup input is not in its table.
 *                  A key is shared, read-only state (upstream's &ProvingKey): several host threads may prove / verify on
 *                  ONE key concurrently, each through its own ctx -- the per-call workspace lives with the (key, ctx) pair
 *                  inside the library.  Calls through one ctx are serialised as everywhere else.
 *                  bzh_pk_free and bzh_pk_set_quotient_module change what those calls run on (workspaces of every ctx, the
 *                  loaded code object): both return BZH_E_ARG, leaving the key as it was, while a bzh_prove_batch /
 *                  bzh_verify_batch on the key is still running on any ctx, and synchronise the whole device before they
 *                  release anything.  Join the proving threads first. */
typedef struct bzh_pk bzh_pk;
/* THE VERIFYING-KEY DIGEST.  The first thing upstream's create_proof and verify_proof absorb is pk.get_vk().hash_into(transcript)
 * (halo2_proofs 0.2.0 plonk.rs, UPSTREAM; the keys come from keygen_vk / keygen_pk at benches/shot.rs:59-61,
 * benches/board.rs:52-54,80-86, src/circuits/shot.rs:915-940, src/circuits/board.rs:907-932): one scalar,
 *     Fp::from_bytes_wide( Blake2b-512(personal = "Halo2-Verify-Key", (s.len() as u64).to_le_bytes() || s) ),
 *     s = format!("{:?}", vk.pinned())
 * -- a digest of Rust's Debug text of the pinned key (moduli, domain, every gate's expression tree as upstream's
 * Expression enum prints it, query lists, fixed + permutation commitments).  That text cannot be produced outside the
 * crate (the tree SHAPES of halo2_gadgets' 19 gates are upstream source, absent here: SURVEY F2), so the digest is an
 * INPUT of this boundary, not something the library derives:
 *   - a circuit blob carries it as 32 canonical little-endian bytes at BZH_CIRCUIT_BLOB_VK_REPR_OFFSET;
 *   - bzh_circuit_create fills in BZH_VK_REPR_PLACEHOLDER (0x1234); proofs made with it verify here (bzh_verify_batch
 *     absorbs the same value) but are REJECTED by the reference's verify_proof, and vice versa;
 *   - bzh_circuit_set_vk_repr installs the real one before bzh_circuit_blob / bzh_pk_create; the Rust-side shim obtains it
 *     once per key (INTEGRATION.md "The verifying-key digest": a 6-line capture transcript around vk.hash_into), or hands
 *     the Debug text to bzh_vk_digest;
 *   - bzh_circuit_vk_repr / bzh_pk_vk_repr report what a circuit / key carries, and whether it is still the placeholder.
 * BZH_E_RANGE: a repr that is not a canonical Fp element (Fp::from_repr(..) is None upstream). */
#define BZH_CIRCUIT_BLOB_VK_REPR_OFFSET 24
#define BZH_VK_REPR_PLACEHOLDER 0x1234
int bzh_vk_digest(const char* pinned_debug, size_t len, uint8_t* out_repr32);
int bzh_pk_vk_repr(const bzh_pk* pk, uint8_t* out_repr32, int* is_placeholder);
int bzh_pk_create(bzh_ctx* ctx, const bzh_bases* srs, const uint8_t* circuit, size_t circuit_len, bzh_pk** out);
int bzh_pk_free(bzh_ctx* ctx, bzh_pk* pk);
/* Params::commit_lagrange: with the table (g_lagrange | u | w) of bzh_params_create set, the instance, advice, permuted-lookup and
 * grand-product columns are committed in the Lagrange basis, as upstream does -- the same group elements (hence the same proof
 * bytes) as committing their coefficients to g, but sparse / small witness columns then cost the MSM almost nothing.
 * NULL returns to coefficient-basis commitments.  The table must outlive the key. */
int bzh_pk_set_lagrange(bzh_pk* pk, const bzh_bases* g_lagrange);
/* the quotient's evaluator program (compiled on the host at bzh_pk_create): instructions, field multiplications per
 * extended-domain row, LDS slots, proof-independent subexpressions hoisted into key-owned coset columns */
int bzh_pk_quotient_stats(bzh_pk* pk, uint32_t* ops, uint32_t* multiplications, uint32_t* lds_slots, uint32_t* hoisted_columns);
/* The quotient evaluator as compiled code.  The program depends on the circuit only (not on k, the SRS or a witness), so for
 * the reference's two circuits (ShotCircuit, BoardCircuit) the kernels are generated when the library is BUILT
 * (csrc/gen_quotient.cpp -> quotient_builtin.hip, linked into libbzh2.so) and picked up by bzh_pk_create through the program's
 * hash: no compiler is needed where the library runs.  For any other circuit bzh_pk_quotient_source returns the key's program
 * as straight-line HIP source (*len = its length, copied NUL-terminated into buf when buf != NULL): compile it for gfx950
 * against csrc/field.cuh (`hipcc -O3 -std=c++17 --offload-arch=gfx950 --genco -I <csrc>`, ~5 s; or hiprtc) and hand the code
 * object to bzh_pk_set_quotient_module, which selects it (BoardCircuit, 64 x 2^17 rows: 35.5 ms against the interpreter's 49.5 ms;
 * same proof bytes).  A module generated from another program is refused (BZH_E_ARG: it carries the program's hash); NULL / 0
 * unloads the module and returns to the key's default (the builtin kernel if there is one, else the interpreter).
 * bzh_pk_quotient_select picks explicitly (BZH_E_RANGE when that flavour is not available for this key);
 * bzh_pk_quotient_selected reports the current choice and whether a builtin kernel exists.  Environment: BZH_QUOTIENT=interp
 * makes the interpreter every new key's default. */
typedef enum { BZH_QUOTIENT_INTERPRETER = 0, BZH_QUOTIENT_BUILTIN = 1, BZH_QUOTIENT_MODULE = 2 } bzh_quotient_flavour;
int bzh_pk_quotient_source(bzh_pk* pk, char* buf, size_t cap, size_t* len);
int bzh_pk_set_quotient_module(bzh_ctx* ctx, bzh_pk* pk, const void* code_object, size_t len);
int bzh_pk_quotient_select(bzh_pk* pk, int flavour);
int bzh_pk_quotient_selected(bzh_pk* pk, int* flavour, int* builtin_available);
/* Host only (no ctx, no GPU): the quotient program of a circuit blob as the source text of a BUILTIN kernel (namespace
 * bzh_q_<hash> with the kernel bzh_quotient_<hash> and a host function `launch`), and the program hash.  This is what the
 * build-time generator calls; BZH_E_RANGE if the circuit does not fit the evaluator (such circuits use the VM v1 fold). */
int bzh_quotient_source_for_circuit(int curve, const uint8_t* circuit, size_t circuit_len, char* buf, size_t cap, size_t* len,
                                    uint64_t* program_hash);
/* Host only: the terms of a circuit's quotient numerator (gate constraints with their compressed selectors, permutation and
 * lookup argument terms, in protocol order) by polynomial degree -- polys[d] terms of degree d (in units of n - 1, d < 16) and
 * muls[d] field multiplications in their expression trees.  A term of degree d vanishes on the 2^k-row domain by itself, so its
 * share of h(X) could be computed from (d - 1) n evaluations instead of the extended domain's; the histogram says what that
 * would buy (halo2's selector compression pads nearly every gate of the reference's circuits to degree
 * 7 - 9, so for them it buys ~3 %). */
int bzh_quotient_degree_histogram(int curve, const uint8_t* circuit, size_t circuit_len, uint32_t* polys16, uint32_t* muls16);
/* launcher of one builtin kernel, and the table the generated file exports (used inside the library) */
typedef void (*bzh_quotient_launch_fn)(unsigned grid_x, unsigned grid_y, void* hip_stream, const uint32_t* const* cols, const size_t* strides,
                                       const uint32_t* consts, size_t const_stride, size_t size, uint32_t* out);
/* launch29 (may be NULL): the same program in unsaturated 9 x 29-bit limbs (csrc/fe29.cuh: 188-instruction products without carry
 * instructions).  Its columns are fe29 planes (9 * size words per column: limbs 0-3 | limbs 4-7 | limb 8), `strides` in words
 * between proofs, constants 12 words apart (9 used), `out` saturated as for `launch`. */
typedef struct { uint64_t program_hash; bzh_quotient_launch_fn launch; const char* name; bzh_quotient_launch_fn launch29; } bzh_builtin_quotient;
const bzh_builtin_quotient* bzh_builtin_quotients(size_t* count);
int bzh_pk_info(const bzh_pk* pk, size_t* rng_bytes_per_proof, size_t* max_proof_bytes, uint32_t* num_advice, uint32_t* n_rows,
                uint32_t* usable_rows);
/* bzh_verify_batch  plonk::verify_proof (SingleVerifier; benches/board.rs:80-86) for `batch` proofs of the key's circuit:
 *                  results[b] = 1 if proof b verifies against instances b, else 0 (malformed proofs included).
 *                  g0_u_w: G_0, U, W of the SRS as 3 affine canonical points; they are checked against the SRS the key
 *                  was built on (BZH_E_ARG if they differ: a stale copy would otherwise reject every valid proof). */
int bzh_verify_batch(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* instances, size_t instance_rows, const uint8_t* proofs,
                     size_t proof_stride, const size_t* proof_lens, const uint64_t* g0_u_w, int* results);
int bzh_prove_batch(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                    size_t instance_rows, const uint8_t* rng, size_t rng_stride, uint8_t* proofs, size_t proof_stride,
                    size_t* proof_lens);

/* create_proof with the randomness drawn inside the library, as the reference does from OsRng (benches/shot.rs:68): proof b's
 * stream is ChaCha20 keyed by seeds[b] (32 bytes; 64-bit block counter from 0, zero nonce), block i being the i-th 64-byte draw
 * (ff::Field::random), expanded on the device -- no rng_bytes_per_proof (2 MB at k = 14) to generate and upload per proof.
 * The proofs are exactly those of bzh_prove_batch fed with bzh_rng_expand(seeds[b], 0, rng_bytes_per_proof / 64, ..).
 * SECURITY: every blinding factor of proof b is a function of seeds[b]; the seeds (and the bytes given to bzh_prove_batch)
 * MUST come from a cryptographically secure generator (getrandom(2), OsRng, os.urandom), fresh per proof -- a predictable or
 * reused seed voids zero-knowledge.  Fixed seeds are for tests and benchmarks only. */
int bzh_prove_batch_seeded(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                           size_t instance_rows, const uint8_t* seeds, uint8_t* proofs, size_t proof_stride, size_t* proof_lens);
/* host: draws [first_draw, first_draw + draws) of the stream of `seed`, 64 bytes each, into out */
int bzh_rng_expand(const uint8_t* seed, uint64_t first_draw, size_t draws, uint8_t* out);

/* ---- Params::new (halo2_proofs poly::commitment::Params; benches/shot.rs:58, benches/board.rs:51, and on every call of
 * the wasm exports, src/wasm/circuit_wasm.rs:57,97,145,180) -------------------------------------------------------
 * A pure function of k: g[i] = hash_to_curve("Halo2-Parameters")(0u8 || i as u32 LE), g_lagrange = inverse group FFT of
 * g, w = hash_to_curve(..)([1]), u = hash_to_curve(..)([2]) on Vesta.
 *   bzh_hash_to_curve      pasta_curves' CurveExt::hash_to_curve(domain_prefix)(msg) on Pallas / Vesta (host): the reference's
 *                          own call site is src/utils/pedersen.rs:19-21 ("battlezips:hash2curve", b"v" / b"r").
 *   bzh_params_generators  g (n x 8 canonical limbs), w, u on the host (threads over i; 0 = auto).
 *   bzh_group_ifft         g -> g_lagrange on the device (n/2 log n + n point-by-scalar multiplications).
 *   bzh_params_create      both of the above, cached on disk keyed by (curve, k) under cache_dir (NULL: next to the library
 *                          or $BZH_CACHE_DIR; "": no cache), then the two commitment-base tables (g | u | w) and
 *                          (g_lagrange | u | w) uploaded with fixed-base window tables: what bzh_pk_create /
 *                          bzh_pk_set_lagrange take.  The params own the tables (bzh_params_free releases them). */
typedef struct bzh_params bzh_params;
int bzh_hash_to_curve(int curve, const char* domain_prefix, const uint8_t* msg, size_t len, uint64_t* out_xy);
int bzh_params_generators(unsigned k, uint64_t* g_xy, uint64_t* w_xy, uint64_t* u_xy, unsigned threads);
int bzh_group_ifft(bzh_ctx* ctx, int curve, const uint64_t* g_xy, unsigned k, uint64_t* out_xy);
int bzh_params_create(bzh_ctx* ctx, unsigned k, const char* cache_dir, int window_bits, bzh_params** out);
int bzh_params_free(bzh_ctx* ctx, bzh_params* p);
int bzh_params_bases(const bzh_params* p, bzh_bases** g, bzh_bases** g_lagrange);
int bzh_params_points(const bzh_params* p, uint64_t* g_xy, uint64_t* g_lagrange_xy, uint64_t* w_xy, uint64_t* u_xy, int* from_cache);

/* ---- circuits: the reference's ShotCircuit / BoardCircuit as data + their witness synthesis -------------------------
 * The reference-side interface this replaces is `impl Circuit<pallas::Base> for {ShotCircuit, BoardCircuit}`
 * (src/circuits/shot.rs:22-53, src/circuits/board.rs:21-51): `configure` -> ShotChip::configure / BoardChip::configure
 * (src/chips/shot.rs:179-297, src/chips/board.rs:194-321) and `synthesize` -> ShotChip::synthesize / BoardChip::synthesize
 * (src/chips/shot.rs:308-354, src/chips/board.rs:331-363), with every chip under them (src/chips/{bitify,placement,
 * transpose,pedersen}.rs and halo2_gadgets' EccChip), laid out by SimpleFloorPlanner.  Host C++, no device needed
 * except for BZH_MEM_DEVICE output.
 *   bzh_circuit_create   configure + the keygen synthesis at 2^k rows (Shot: k >= 11, Board: k >= 12 -- the reference's
 *                        sizes, benches/shot.rs:22, benches/board.rs:22; BZH_E_RANGE if the circuit does not fit) +
 *                        selector compression.  `bits`: B of the two bitify test circuits (src/chips/bitify.rs:255-403),
 *                        ignored otherwise.
 *   bzh_circuit_blob     the serialised constraint system + fixed assignment for bzh_pk_create ("BZC2", csrc/prove.hip);
 *                        out = NULL only reports the length.
 *   bzh_circuit_describe JSON: column counts, gates (names, constraint names, queried cells), regions (name, rows,
 *                        columns), permutation columns, query lists -- what dev::MockProver reports failures against.
 *   bzh_synthesize_shot / bzh_synthesize_board
 *                        `batch` witnesses: fills advice = batch x num_advice x 2^k elements (`form`; BZH_MEM_HOST, or
 *                        BZH_MEM_DEVICE + BZH_FORM_MONTGOMERY: staged compactly through pinned memory and expanded on the
 *                        ctx's stream) -- exactly the tensor bzh_prove_batch consumes -- and instances = batch x
 *                        {4, 2} canonical public inputs (Shot: commitment x, y, shot, hit: src/circuits/shot.rs:125-130;
 *                        Board: commitment x, y: src/circuits/board.rs:113-118).  Inputs are the constructor arguments of
 *                        ShotCircuit::new / BoardCircuit::new: 256-bit BinaryValues as 4 limbs, trapdoors as canonical
 *                        Pallas scalars.  `threads` host threads (0 = auto).  BZH_E_RANGE: an input the reference would
 *                        panic on (H and V share a bit: src/utils/binary.rs:97-108; non-canonical trapdoor).
 *   bzh_board_witness    Board::from(&Deck::from(ships)).{witness, state}(options): ships = 5 x (x, y, z), x < 0 = None;
 *                        options = 5 WitnessOption values (src/utils/ship.rs:315-331) or NULL.  Anything but NULL / all
 *                        BZH_WITNESS_DEFAULT is the reference's TEST-ONLY fault injection (malicious ship commitments for its
 *                        negative MockProver tests): a negative-test aid, never a production input;
 *                        out: 10 ship commitments [H5, V5, H4, V4, H3a, V3a, H3b, V3b, H2, V2] and the board state.
 *   bzh_shot_serialize   shot::serialize (src/utils/shot.rs:12-19).
 *   bzh_pedersen_commit_host   pedersen_commit (src/utils/pedersen.rs:17-28) on the host, from the circuit's window tables.
 *   bzh_fixed_base_tables      Z / U / Lagrange tables of BoardCommitV (0) and BoardCommitR (1), derived from the generators
 *                        (src/utils/constants/fixed_bases/board_commit_{v,r}.rs). */
typedef struct bzh_circuit bzh_circuit;
typedef enum { BZH_CIRCUIT_SHOT = 0, BZH_CIRCUIT_BOARD = 1, BZH_CIRCUIT_NUM2BITS_TEST = 2, BZH_CIRCUIT_BITS2NUM_TEST = 3 } bzh_circuit_kind;
typedef enum {
    BZH_WITNESS_DEFAULT = 0,
    BZH_WITNESS_DUAL_PLACEMENT = 1,
    BZH_WITNESS_NONCONSECUTIVE = 2,
    BZH_WITNESS_EXTRA_BIT = 3,
    BZH_WITNESS_OVERSIZED = 4,
    BZH_WITNESS_UNDERSIZED = 5
} bzh_witness_option;
int bzh_circuit_create(int kind, unsigned k, unsigned bits, bzh_circuit** out);
int bzh_circuit_free(bzh_circuit* c);
const char* bzh_circuit_last_error(void);
int bzh_circuit_blob(const bzh_circuit* c, uint8_t* out, size_t cap, size_t* len);
int bzh_circuit_describe(const bzh_circuit* c, char* out, size_t cap, size_t* len);
int bzh_circuit_info(const bzh_circuit* c, uint32_t* num_advice, uint32_t* num_instance_rows, uint32_t* n_rows, uint32_t* rows_used,
                     uint32_t* num_gates, uint32_t* num_regions);
/* the verifying-key digest of the blob this circuit hands out (see "THE VERIFYING-KEY DIGEST" above): set it BEFORE
 * bzh_circuit_blob / bzh_pk_create; *is_placeholder = 1 until it has been set */
int bzh_circuit_set_vk_repr(bzh_circuit* c, const uint8_t* repr32);
int bzh_circuit_vk_repr(const bzh_circuit* c, uint8_t* out_repr32, int* is_placeholder);
int bzh_synthesize_shot(bzh_ctx* ctx, const bzh_circuit* c, size_t batch, const uint64_t* boards, const uint64_t* trapdoors, const uint64_t* shots,
                        const uint64_t* hits, uint64_t* advice, int form, int mem, uint64_t* instances, unsigned threads);
int bzh_synthesize_board(bzh_ctx* ctx, const bzh_circuit* c, size_t batch, const uint64_t* ship_commitments, const uint64_t* boards,
                         const uint64_t* trapdoors, uint64_t* advice, int form, int mem, uint64_t* instances, unsigned threads);
int bzh_synthesize_bitify_test(const bzh_circuit* c, const uint64_t* value, const uint64_t* binary, uint64_t* advice);
int bzh_board_witness(const int8_t* ships, const int32_t* options, uint64_t* ship_commitments, uint64_t* state);
int bzh_shot_serialize(const uint8_t* xs, const uint8_t* ys, size_t count, uint64_t* out);
int bzh_pedersen_commit_host(const uint64_t* message, const uint64_t* trapdoor, uint64_t* out_xy);
/* The same commitment for n (message, trapdoor) pairs in ONE launch on the device (the reference calls
 * pedersen_commit once per proof from every frontend and a second time inside ShotChip::synthesize, src/chips/shot.rs:319,
 * re-hashing V and R each time).  messages / trapdoors: n x 4 canonical limbs (host) -- the message is an Fp value re-read as a
 * Pallas scalar through its repr, as pedersen.rs:23-24 does; out_xy: n affine canonical points x || y, (0, 0) = identity.
 * V and R live in a ctx-owned direct-lookup table (d * 2^(8w) * G for 32 signed 8-bit windows, 512 KB, built on the first
 * call): 64 mixed additions + one inversion per commitment, one lane each.  BZH_E_RANGE: a repr that is not a canonical
 * Pallas scalar (from_repr(..).unwrap() panics upstream).  Returns when out_xy is filled. */
int bzh_pedersen_commit_batch(bzh_ctx* ctx, const uint64_t* messages, const uint64_t* trapdoors, size_t n, uint64_t* out_xy);
int bzh_fixed_base_tables(int base, uint64_t* z, uint64_t* u, uint64_t* lagrange);

/* ---- host helpers (CPU, no device needed): what `.to_affine()` / `to_bytes()`
 * do on the Rust side; used by tests and benches to compare canonical bytes. */
int bzh_jacobian_to_affine(int curve, const uint64_t* xyz, size_t n, int form, uint64_t* out_xy);
/* sum of n Jacobian points (host): the combine step of one MSM whose points are split over several GPUs -- every rank
 * holds N/R points, returns one 96-byte partial, the partials are all-gathered and added locally. */
int bzh_jacobian_sum(int curve, const uint64_t* xyz, size_t n, int form, uint64_t* out_xyz);
/* pasta_curves to_bytes: x little-endian, bit 255 = parity of canonical y, identity = 32 zero bytes.
 * xy in `form`; out: n * 32 bytes. */
int bzh_affine_compress(int curve, const uint64_t* xy, size_t n, int form, uint8_t* out32);
/* root of unity of order 2^log_n used by halo2's EvaluationDomain for this field
 * (ROOT_OF_UNITY^(2^(S-log_n))); out: 4 limbs in `form`. */
int bzh_field_omega(int field, unsigned log_n, int form, uint64_t* out);

/* ---- proof / IO record (host) ---------------------------------------------
 * The record the reference's wasm frontend hands to JavaScript, src/wasm/circuit_wasm.rs:27-31:
 *     struct BattleZipsWASM { commitment: Vec<[u8; 32]>, proof: Vec<u8> }
 * commitment = the public inputs (Board: commit.x, commit.y; Shot: commit.x, commit.y, shot, hit) as
 * BinaryValue::from_fp(fp).to_repr() (:75-83, :164-167), proof = Blake2bWrite::finalize().  Two forms:
 *   - the serde JSON text {"commitment":[[32 numbers],...],"proof":[numbers]} (bzh_record_to_json / _from_json), and
 *   - a fixed-stride binary record: what the multi-GPU gather carries (equal-sized records, one all_gather) and what
 *     a batch client stores per proof:
 *         u32 proof_len | u8 n_inputs | u8 kind | u16 0 | u32 index | u32 0 | u8 inputs[4][32] | u8 proof[proof_stride]
 *     `kind` / `index` are the caller's tags (e.g. 0 = Board, 1 = Shot; position in the batch); they are not part of the
 *     JSON form.
 * Reading back (verify_board / verify_shot, :86-116) goes through BinaryValue::from_repr(bin).to_fp(): a public input
 * that is not a canonical Fp element is refused -- BZH_E_RANGE here, from encode, decode and from_json alike.
 * Malformed JSON is BZH_E_ARG; a proof longer than the record's stride, more than 4 inputs, or a corrupt header is
 * BZH_E_RANGE. */
#define BZH_RECORD_HEADER_BYTES 144
#define BZH_RECORD_MAX_INPUTS 4
/* bytes per record for proofs of up to proof_stride bytes (bzh_pk_info's max_proof_bytes) */
size_t bzh_record_stride(size_t proof_stride);
/* public_inputs: n_inputs x 4 canonical limbs */
int bzh_record_encode(const uint64_t* public_inputs, size_t n_inputs, const uint8_t* proof, size_t proof_len, uint32_t kind, uint32_t index,
                      uint8_t* record, size_t record_stride);
/* any out pointer may be NULL; *proof points into `record` */
int bzh_record_decode(const uint8_t* record, size_t record_stride, uint64_t* public_inputs, size_t* n_inputs, const uint8_t** proof,
                      size_t* proof_len, uint32_t* kind, uint32_t* index);
/* out == NULL: size query (*len = strlen of the text); otherwise cap must hold *len + 1 */
int bzh_record_to_json(const uint8_t* record, size_t record_stride, char* out, size_t cap, size_t* len);
int bzh_record_from_json(const char* text, size_t text_len, uint32_t kind, uint32_t index, uint8_t* record, size_t record_stride);

#ifdef __cplusplus
}
#endif
#endif /* BZH2_H */
