// The reference's benches/shot.rs (and the `production` test, src/circuits/shot.rs:880-941) as a stand-alone C++ client of
// the C ABI (include/bzh2.h): no Python, no torch, no case files.
//
//   benches/shot.rs:24-47   Board::from(&Deck::from(..)), serialize::<1>(..), pedersen_commit   -> bzh_board_witness, bzh_shot_serialize
//   benches/shot.rs:58      Params::<vesta::Affine>::new(K)                                      -> bzh_params_create
//   benches/shot.rs:60-61   keygen_vk / keygen_pk (configure + keygen synthesize)                -> bzh_circuit_create + bzh_pk_create
//   benches/shot.rs:64-71   b.iter(|| create_proof(&params, &pk, &[circuit], &[&[&public_inputs]], OsRng, &mut transcript))
//                                                                                                -> bzh_synthesize_shot + bzh_prove_batch
//   benches/shot.rs:80-86   verify_proof (commented out there; benches/board.rs:80-86)           -> bzh_verify_batch
// This is the call sequence a Rust shim inside the reference would make.
//
//   g++ -O2 -std=c++17 -I include examples/shot_prover.cpp -o examples/shot_prover -L battlezips-halo2_amd -lbzh2 \
//       -Wl,-rpath,$PWD/battlezips-halo2_amd -lpthread
//   examples/shot_prover [batch] [steps]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <sys/random.h>

#include "bzh2.h"

#define CHECK(expr)                                                                              \
    do {                                                                                         \
        int rc__ = (expr);                                                                       \
        if (rc__ != BZH_OK) {                                                                    \
            fprintf(stderr, "%s failed: %s (%d)\n", #expr, bzh_strerror(rc__), rc__);            \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

// OsRng: getrandom(2); the fixed generator only on request
static void fill_random(uint8_t* out, size_t n, bool fixed_seed, std::mt19937_64& gen) {
    if (!fixed_seed) {
        size_t got = 0;
        while (got < n) {
            const ssize_t r = getrandom(out + got, n - got, 0);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got == n) return;
        fprintf(stderr, "getrandom failed\n");
        exit(1);
    }
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)gen();
}

int main(int argc, char** argv) {
    const size_t batch = argc > 1 ? (size_t)atoi(argv[1]) : 16;
    const int steps = argc > 2 ? atoi(argv[2]) : 3;
    const unsigned K = 11;                                         // benches/shot.rs:22
    bzh_ctx* ctx = nullptr;
    CHECK(bzh_ctx_create(0, &ctx));

    // board pattern #1 and a shot at (3, 5): a hit (src/circuits/shot.rs:102-112; benches/shot.rs shoots (3, 3) and asserts a
    // miss on an occupied cell, which cannot verify -- SURVEY F7)
    const int8_t ships[15] = {3, 3, 1, 5, 4, 0, 0, 1, 0, 0, 5, 1, 6, 1, 0};
    uint64_t ship_commitments[40], board_state[4], shot[4];
    CHECK(bzh_board_witness(ships, nullptr, ship_commitments, board_state));
    const uint8_t sx = 3, sy = 5;
    CHECK(bzh_shot_serialize(&sx, &sy, 1, shot));

    bzh_params* params = nullptr;
    CHECK(bzh_params_create(ctx, K, nullptr, 0, &params));        // Params::new(K), cached on disk
    bzh_bases *g = nullptr, *g_lagrange = nullptr;
    CHECK(bzh_params_bases(params, &g, &g_lagrange));

    bzh_circuit* circuit = nullptr;
    CHECK(bzh_circuit_create(BZH_CIRCUIT_SHOT, K, 0, &circuit));  // ShotChip::configure + keygen synthesize
    // The verifying-key digest (what create_proof / verify_proof absorb first: vk.hash_into) is an INPUT of the boundary: a Rust
    // shim hands over Fp::to_repr of the scalar it captured from keygen_vk's key; here it can come from the
    // environment as 64 hex digits (little-endian repr).  Without it the library's placeholder is used: such proofs verify with
    // bzh_verify_batch, not with the reference's verify_proof.
    if (const char* hex = getenv("BZH_EXAMPLE_VK_REPR")) {
        uint8_t repr[32];
        bool ok = strlen(hex) == 64;
        for (int i = 0; i < 32 && ok; i++) {
            unsigned v = 0;
            ok = sscanf(hex + 2 * i, "%2x", &v) == 1;
            repr[i] = (uint8_t)v;
        }
        if (!ok) {
            fprintf(stderr, "BZH_EXAMPLE_VK_REPR: 64 hex digits expected\n");
            return 1;
        }
        CHECK(bzh_circuit_set_vk_repr(circuit, repr));
    }
    int vk_placeholder = 0;
    CHECK(bzh_circuit_vk_repr(circuit, nullptr, &vk_placeholder));
    size_t blob_len = 0;
    CHECK(bzh_circuit_blob(circuit, nullptr, 0, &blob_len));
    std::vector<uint8_t> blob(blob_len);
    CHECK(bzh_circuit_blob(circuit, blob.data(), blob.size(), &blob_len));
    bzh_pk* pk = nullptr;
    CHECK(bzh_pk_create(ctx, g, blob.data(), blob.size(), &pk));  // keygen_pk
    CHECK(bzh_pk_set_lagrange(pk, g_lagrange));

    size_t rng_bytes = 0, max_proof = 0;
    uint32_t num_advice = 0, n_rows = 0, usable = 0;
    CHECK(bzh_pk_info(pk, &rng_bytes, &max_proof, &num_advice, &n_rows, &usable));

    // `batch` circuits: same board and shot, a fresh trapdoor each (pallas::Scalar::random(&mut OsRng)).  Seeds and
    // trapdoors MUST come from the OS generator: a predictable seed makes every blind of the proof predictable and breaks
    // zero-knowledge.  BZH_EXAMPLE_FIXED_SEED=1 (benchmarks / reproducible runs only) uses mt19937_64(42) instead.
    const bool fixed_seed = getenv("BZH_EXAMPLE_FIXED_SEED") != nullptr;
    std::mt19937_64 gen(42);
    std::vector<uint64_t> boards(4 * batch), shots(4 * batch), hits(4 * batch, 0), trapdoors(4 * batch);
    for (size_t b = 0; b < batch; b++) {
        memcpy(&boards[4 * b], board_state, 32);
        memcpy(&shots[4 * b], shot, 32);
        hits[4 * b] = 1;
        fill_random((uint8_t*)&trapdoors[4 * b], 32, fixed_seed, gen);
        trapdoors[4 * b + 3] &= 0x3fffffffffffffffull;            // < 2^254 < q
    }
    std::vector<uint64_t> advice(batch * num_advice * (size_t)n_rows * 4), instances(batch * 4 * 4);
    std::vector<uint8_t> seeds(batch * 32), proofs(batch * max_proof);   // 32 bytes of OsRng per proof; the library expands them
    std::vector<size_t> lens(batch);
    std::vector<int> ok(batch);
    double best = 1e30;
    // the quotient evaluator: ShotCircuit's kernel is inside libbzh2.so (generated when the library was built) and was picked
    // by bzh_pk_create; BZH_EXAMPLE_NO_CODEGEN=1 selects the interpreter
    if (getenv("BZH_EXAMPLE_NO_CODEGEN")) CHECK(bzh_pk_quotient_select(pk, BZH_QUOTIENT_INTERPRETER));
    int flavour = 0;
    CHECK(bzh_pk_quotient_selected(pk, &flavour, nullptr));
    for (int s = 0; s < steps; s++) {
        fill_random(seeds.data(), seeds.size(), fixed_seed, gen);   // OsRng: every blinding factor of a proof derives from its seed
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(bzh_synthesize_shot(ctx, circuit, batch, boards.data(), trapdoors.data(), shots.data(), hits.data(), advice.data(),
                                  BZH_FORM_MONTGOMERY, BZH_MEM_HOST, instances.data(), 0));
        CHECK(bzh_prove_batch_seeded(ctx, pk, batch, advice.data(), BZH_FORM_MONTGOMERY, BZH_MEM_HOST, instances.data(), 4, seeds.data(),
                                     proofs.data(), max_proof, lens.data()));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best) best = ms;
    }
    uint64_t g0_u_w[24];                                          // G_0, U, W for the verifier (host copies of the SRS)
    {
        std::vector<uint64_t> gxy((size_t)8 << K);
        CHECK(bzh_params_points(params, gxy.data(), nullptr, g0_u_w + 16, g0_u_w + 8, nullptr));
        memcpy(g0_u_w, gxy.data(), 64);
    }
    CHECK(bzh_verify_batch(ctx, pk, batch, instances.data(), 4, proofs.data(), max_proof, lens.data(), g0_u_w, ok.data()));
    size_t accepted = 0;
    for (int v : ok) accepted += v != 0;
    uint64_t h = 1469598103934665603ull;                          // FNV-1a of the last batch's proofs
    for (size_t b = 0; b < batch; b++)
        for (size_t i = 0; i < lens[b]; i++) h = (h ^ proofs[b * max_proof + i]) * 1099511628211ull;
    printf("{\"circuit\": \"ShotCircuit k=11\", \"batch\": %zu, \"best_ms\": %.2f, \"proofs_per_s\": %.1f, \"proof_bytes\": %zu, \"verified\": %zu, "
           "\"quotient\": \"%s\", \"vk_digest\": \"%s\", \"fnv1a\": \"%016llx\"}\n", batch, best, batch / best * 1e3, lens[0], accepted,
           flavour == BZH_QUOTIENT_BUILTIN ? "builtin kernel" : (flavour == BZH_QUOTIENT_MODULE ? "module" : "interpreted"),
           vk_placeholder ? "placeholder (BZH_EXAMPLE_VK_REPR not given)" : "caller's", (unsigned long long)h);
    // what the reference's wasm frontend returns per proof (src/wasm/circuit_wasm.rs:27-31,164-170): {commitment, proof}
    {
        const size_t stride = bzh_record_stride(max_proof);
        std::vector<uint8_t> rec(stride);
        CHECK(bzh_record_encode(instances.data(), 4, proofs.data(), lens[0], /*kind: Shot*/ 1, 0, rec.data(), stride));
        size_t jl = 0;
        CHECK(bzh_record_to_json(rec.data(), stride, nullptr, 0, &jl));
        std::string json(jl + 1, '\0');
        CHECK(bzh_record_to_json(rec.data(), stride, &json[0], jl + 1, &jl));
        std::vector<uint8_t> back(stride);
        CHECK(bzh_record_from_json(json.c_str(), jl, 1, 0, back.data(), stride));   // canonical check of verify_shot (:86-116)
        printf("{\"record_json_bytes\": %zu, \"round_trip\": %s}\n", jl, back == rec ? "true" : "false");
    }
    bzh_pk_free(ctx, pk);
    bzh_circuit_free(circuit);
    bzh_params_free(ctx, params);
    bzh_ctx_destroy(ctx);
    return accepted == batch ? 0 : 2;
}
