// GENERATOR constants of the two fixed bases of the BattleZips Pedersen commitment: hash_to_curve("battlezips:hash2curve")(b"v" /
// b"r") on Pallas, canonical little-endian limbs x || y (src/utils/constants/fixed_bases/board_commit_v.rs:5-14,
// board_commit_r.rs:5-14; their `generator` tests :2941-2948 -- reproduced by bzh_hash_to_curve in tests/test_params_cpu.py).
#pragma once
#include <stdint.h>

namespace bzh {
static const uint64_t PEDERSEN_GEN_V[8] = {0x0aaf6299a6692ca4ull, 0xbd7d655cae1385d7ull, 0xaa3fc3f9268467a3ull, 0x1e2542d216c42158ull,
                                           0x37a9d7aa880f14b2ull, 0xe705a08374ba472full, 0x0a26f1bc8cffd318ull, 0x32c5c94a039386f8ull};
static const uint64_t PEDERSEN_GEN_R[8] = {0xdf6e95bb730e5277ull, 0x1693cb90e5c65c55ull, 0xcfb7d2c81d8bc40cull, 0x1c332c6fa1a9c3d7ull,
                                           0x04962380847950b8ull, 0xccc5ff9ca952893full, 0x75d6a964af190d95ull, 0x0f9890742e8ad0e5ull};
}  // namespace bzh
