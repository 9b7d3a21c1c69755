// Game / witness marshalling of the reference, host C++:
//   BinaryValue            src/utils/binary.rs:16-109     256-bit little-endian bit array
//   Ship / WitnessOption   src/utils/ship.rs:147-212, 220-311, 315-331
//   Deck                   src/utils/deck.rs:53-71
//   Board::{state,witness} src/utils/board.rs:77-120
//   serialize (shot)       src/utils/shot.rs:12-19
// Same names, argument meaning and failure behaviour (a panic upstream is a GameError here).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "hostfield.hpp"

namespace bzc {

static constexpr int BOARD_SIZE = 100;  // src/utils/board.rs:12

struct GameError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct BinaryValue {
    uint64_t w[4] = {0, 0, 0, 0};
    static BinaryValue empty() { return BinaryValue(); }
    static BinaryValue from_u8(uint8_t v) {
        BinaryValue b;
        b.w[0] = v;
        return b;
    }
    static BinaryValue from_limbs(const uint64_t* l) {
        BinaryValue b;
        memcpy(b.w, l, 32);
        return b;
    }
    bool bit(int i) const { return (w[i / 64] >> (i % 64)) & 1; }
    void set(int i, bool v) {
        if (i < 0 || i >= 256) throw GameError("BinaryValue: bit index out of range");
        if (v) {
            w[i / 64] |= (uint64_t)1 << (i % 64);
        } else {
            w[i / 64] &= ~((uint64_t)1 << (i % 64));
        }
    }
    u128 lower_u128() const { return (u128)w[0] | ((u128)w[1] << 64); }
    // to_fp: Fp::from_repr(..).unwrap() -- a non-canonical value is an error upstream
    Fp to_fp() const {
        Fp v;
        if (!Fp::from_limbs(w, &v)) throw GameError("BinaryValue::to_fp: not a canonical Fp representation");
        return v;
    }
    Fp bit_fp(int i) const { return bit(i) ? Fp::one() : Fp::zero(); }
    // zip: OR of the first 100 bits; both set is a panic upstream (src/utils/binary.rs:97-108)
    BinaryValue zip(const BinaryValue& to) const {
        BinaryValue z;
        for (int i = 0; i < BOARD_SIZE; i++) {
            if (bit(i) && to.bit(i)) throw GameError("Cannot zip together bit #" + std::to_string(i));
            z.set(i, bit(i) || to.bit(i));
        }
        return z;
    }
};

enum WitnessOption { WO_DEFAULT = 0, WO_DUAL_PLACEMENT, WO_NONCONSECUTIVE, WO_EXTRA_BIT, WO_OVERSIZED, WO_UNDERSIZED };

inline int ship_length(int ship_type) {  // get_ship_length, src/utils/ship.rs:24-33
    static const int len[5] = {5, 4, 3, 3, 2};
    return (ship_type >= 0 && ship_type < 5) ? len[ship_type] : 0;
}

struct Ship {
    int ship_type;
    int x, y;
    bool z;
    std::vector<int> coordinates(bool transpose) const {
        std::vector<int> out;
        for (int i = 0; i < ship_length(ship_type); i++) {
            const int x_i = z ? x : x + i, y_i = z ? y + i : y;
            const int xs = (transpose && z) ? x_i * 10 : x_i;
            const int ys = (transpose && z) ? y_i : y_i * 10;
            out.push_back(xs + ys);
        }
        return out;
    }
    BinaryValue bits(bool transpose) const {
        BinaryValue b;
        for (int c : coordinates(transpose)) b.set(c, true);
        return b;
    }
    // [H, V] placement commitments, optionally tampered (src/utils/ship.rs:189-311)
    void witness(int utility, BinaryValue out[2]) const {
        out[0] = out[1] = BinaryValue::empty();
        const int target = z ? 1 : 0;
        out[target] = bits(true);
        const std::vector<int> co = coordinates(true);
        const int first = co.front(), last = co.back();
        switch (utility) {
            case WO_DEFAULT: break;
            case WO_DUAL_PLACEMENT:
                out[1 - target].set(first, true);
                out[target].set(first, false);
                break;
            case WO_NONCONSECUTIVE:
                out[target].set(last, false);
                out[target].set(last + 1, true);
                break;
            case WO_EXTRA_BIT: out[target].set(0, true); break;
            case WO_OVERSIZED: out[target].set(last + 1, true); break;
            case WO_UNDERSIZED: out[target].set(last, false); break;
            default: throw GameError("unknown WitnessOption");
        }
    }
};

struct Deck {  // five optional ships, carrier .. destroyer
    bool present[5] = {false, false, false, false, false};
    Ship ships[5];
    void add(int type, int x, int y, bool z) {
        present[type] = true;
        ships[type] = Ship{type, x, y, z};
    }
};

struct Board {
    Deck deck;
    // OR of every H placement and of every V placement re-indexed j -> (j % 10) * 10 + j / 10 (src/utils/board.rs:77-98)
    BinaryValue state(const int utilities[5]) const {
        BinaryValue st;
        for (int i = 0; i < 5; i++) {
            if (!deck.present[i]) continue;
            BinaryValue pl[2];
            deck.ships[i].witness(utilities[i], pl);
            for (int j = 0; j < BOARD_SIZE; j++) {
                if (pl[0].bit(j)) st.set(j, true);
                if (pl[1].bit(j)) st.set(j % 10 * 10 + j / 10, true);
            }
        }
        return st;
    }
    // [H5, V5, H4, V4, H3a, V3a, H3b, V3b, H2, V2] (src/utils/board.rs:107-120)
    void witness(const int utilities[5], BinaryValue out[10]) const {
        for (int i = 0; i < 5; i++) {
            if (!deck.present[i]) {
                out[2 * i] = out[2 * i + 1] = BinaryValue::empty();
            } else {
                deck.ships[i].witness(utilities[i], &out[2 * i]);
            }
        }
    }
};

// serialize::<1>: bit (10 y + x) (src/utils/shot.rs:12-19)
inline BinaryValue serialize_shot(const uint8_t* xs, const uint8_t* ys, int count) {
    BinaryValue b;
    for (int i = 0; i < count; i++) b.set((int)ys[i] * 10 + xs[i], true);
    return b;
}

}  // namespace bzc
