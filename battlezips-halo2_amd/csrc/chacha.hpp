// ChaCha20 block function (D. J. Bernstein; 20 rounds, 64-bit block counter in words 12-13, zero nonce in words 14-15):
// the random stream of bzh_prove_batch_seeded.  Block i of the stream keyed by a proof's 32-byte seed is the i-th 64-byte
// draw of that proof (ff::Field::random reads 64 bytes per scalar), so host code that needs single draws and device code
// that needs rows of them address the same stream by position.  The reference draws from OsRng inside create_proof
// (benches/shot.rs:68); the byte-exact parity tests keep using explicit streams (bzh_prove_batch).
#pragma once
#include <cstdint>
#include <cstring>

namespace bzh {

#if defined(__HIPCC__)
#define BZH_CHACHA_HD __host__ __device__ inline
#else
#define BZH_CHACHA_HD inline
#endif

BZH_CHACHA_HD uint32_t chacha_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define BZH_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = chacha_rotl(d, 16); \
    c += d; b ^= c; b = chacha_rotl(b, 12); \
    a += b; d ^= a; d = chacha_rotl(d, 8);  \
    c += d; b ^= c; b = chacha_rotl(b, 7);

// out: 16 little-endian words = 64 bytes
BZH_CHACHA_HD void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4],      key[5],      key[6],      key[7],      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = s[i];
    for (int r = 0; r < 10; r++) {
        BZH_CHACHA_QR(x[0], x[4], x[8], x[12])
        BZH_CHACHA_QR(x[1], x[5], x[9], x[13])
        BZH_CHACHA_QR(x[2], x[6], x[10], x[14])
        BZH_CHACHA_QR(x[3], x[7], x[11], x[15])
        BZH_CHACHA_QR(x[0], x[5], x[10], x[15])
        BZH_CHACHA_QR(x[1], x[6], x[11], x[12])
        BZH_CHACHA_QR(x[2], x[7], x[8], x[13])
        BZH_CHACHA_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
#undef BZH_CHACHA_QR

}  // namespace bzh
