// Stand-alone C++ client of the C ABI (include/bzh2.h): no Python, no torch.  Loads a case file written by
// examples/export_case.py (SRS, serialised circuit, witness, instances, randomness), then
//   bzh_ctx_create -> bzh_bases_upload + bzh_bases_precompute -> bzh_pk_create -> bzh_prove_batch -> bzh_verify_batch
// on `threads` host threads (one ctx + key each), and prints a checksum of the proofs and the throughput.
// This is the call sequence a Rust shim of the reference would make where it calls keygen_pk / create_proof /
// verify_proof today (benches/shot.rs:58-71, benches/board.rs:51-86); see INTEGRATION.md.
//
//   g++ -O2 -std=c++17 -I include examples/prove_batch.cpp -o examples/prove_batch -L battlezips-halo2_amd -lbzh2 \
//       -Wl,-rpath,$PWD/battlezips-halo2_amd -lpthread
//   examples/prove_batch case.bin [threads] [steps]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "bzh2.h"

struct Case {
    uint32_t k, num_advice, num_instance, instance_rows, batch;
    std::vector<uint64_t> srs;        // (n + 2) x 8: G.., U, W affine canonical
    std::vector<uint8_t> circuit;     // blob for bzh_pk_create
    std::vector<uint64_t> advice;     // batch x num_advice x n x 4 canonical
    std::vector<uint64_t> instances;  // batch x num_instance x instance_rows x 4 canonical
    std::vector<uint8_t> rng;         // batch x rng_stride
    uint64_t rng_stride;
};

static bool read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

static bool load(const char* path, Case& c) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[4];
    uint64_t len[5];
    bool ok = read_exact(f, magic, 4) && !memcmp(magic, "BZX1", 4) && read_exact(f, &c.k, 20) && read_exact(f, &c.rng_stride, 8) &&
              read_exact(f, len, sizeof(len));
    if (ok) {
        c.srs.resize(len[0] / 8);
        c.circuit.resize(len[1]);
        c.advice.resize(len[2] / 8);
        c.instances.resize(len[3] / 8);
        c.rng.resize(len[4]);
        ok = read_exact(f, c.srs.data(), len[0]) && read_exact(f, c.circuit.data(), len[1]) && read_exact(f, c.advice.data(), len[2]) &&
             read_exact(f, c.instances.data(), len[3]) && read_exact(f, c.rng.data(), len[4]);
    }
    fclose(f);
    return ok;
}

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc__ = (call);                                                                  \
        if (rc__) {                                                                         \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc__, bzh_strerror(rc__));            \
            exit(2);                                                                        \
        }                                                                                   \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s case.bin [threads] [steps]\n", argv[0]);
        return 1;
    }
    Case c;
    if (!load(argv[1], c)) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    const int threads = argc > 2 ? atoi(argv[2]) : 1, steps = argc > 3 ? atoi(argv[3]) : 3;
    const size_t n = (size_t)1 << c.k;
    std::vector<std::vector<uint8_t>> proofs(threads);
    std::vector<std::vector<size_t>> lens(threads);
    std::vector<size_t> stride(threads);
    std::vector<int> accepted(threads, 0);
    std::vector<bzh_ctx*> ctxs(threads);
    std::vector<bzh_bases*> srs(threads);
    std::vector<bzh_pk*> pks(threads);
    for (int t = 0; t < threads; t++) {  // setup: one ctx, SRS window table and proving key per host thread
        CHECK(bzh_ctx_create(0, &ctxs[t]));
        CHECK(bzh_bases_upload(ctxs[t], BZH_CURVE_VESTA, c.srs.data(), n + 2, BZH_FORM_CANONICAL, BZH_MEM_HOST, &srs[t]));
        CHECK(bzh_bases_precompute(ctxs[t], srs[t], 0));
        CHECK(bzh_pk_create(ctxs[t], srs[t], c.circuit.data(), c.circuit.size(), &pks[t]));
        size_t need = 0, maxp = 0;
        CHECK(bzh_pk_info(pks[t], &need, &maxp, nullptr, nullptr, nullptr));
        if (need > c.rng_stride) {
            fprintf(stderr, "case holds %llu randomness bytes per proof, the key needs %zu\n", (unsigned long long)c.rng_stride, need);
            return 2;
        }
        stride[t] = maxp;
        proofs[t].assign((size_t)c.batch * maxp, 0);
        lens[t].assign(c.batch, 0);
    }
    auto work = [&](int t, int reps) {
        for (int s = 0; s < reps; s++)
            CHECK(bzh_prove_batch(ctxs[t], pks[t], c.batch, c.advice.data(), BZH_FORM_CANONICAL, BZH_MEM_HOST, c.instances.data(),
                                  c.instance_rows, c.rng.data(), c.rng_stride, proofs[t].data(), stride[t], lens[t].data()));
    };
    work(0, 1);  // warm-up (arena, programs)
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work, t, steps);
    work(0, steps);
    for (auto& x : th) x.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // verify what thread 0 made, with G_0, U, W taken from the case's SRS
    std::vector<uint64_t> g0uw(24);
    memcpy(&g0uw[0], &c.srs[0], 64);
    memcpy(&g0uw[8], &c.srs[n * 8], 64);
    memcpy(&g0uw[16], &c.srs[(n + 1) * 8], 64);
    std::vector<int> res(c.batch, 0);
    CHECK(bzh_verify_batch(ctxs[0], pks[0], c.batch, c.instances.data(), c.instance_rows, proofs[0].data(), stride[0], lens[0].data(),
                           g0uw.data(), res.data()));
    int ok = 0;
    for (int v : res) ok += v;
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the proof bytes, proof by proof
    for (uint32_t b = 0; b < c.batch; b++)
        for (size_t i = 0; i < lens[0][b]; i++) h = (h ^ proofs[0][b * stride[0] + i]) * 1099511628211ull;
    printf("{\"k\": %u, \"batch\": %u, \"threads\": %d, \"steps\": %d, \"proof_bytes\": %zu, \"verified\": %d, \"fnv1a\": \"%016llx\", "
           "\"proofs_per_s\": %.1f}\n",
           c.k, c.batch, threads, steps, lens[0][0], ok, (unsigned long long)h, (double)c.batch * threads * steps / sec);
    for (int t = 0; t < threads; t++) {
        bzh_pk_free(ctxs[t], pks[t]);
        bzh_bases_free(ctxs[t], srs[t]);
        bzh_ctx_destroy(ctxs[t]);
    }
    return ok == (int)c.batch ? 0 : 3;
}
