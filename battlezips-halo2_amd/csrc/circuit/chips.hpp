// The reference's own chips and circuits, restated gate for gate and cell for cell:
//   Num2BitsChip / Bits2NumChip   src/chips/bitify.rs:45-139, 142-234
//   PlacementChip<S>              src/chips/placement.rs:107-265 (5 gates), :267-369 (3 regions), :380-419 (trace)
//   TransposeChip                 src/chips/transpose.rs:46-88, 99-146
//   PedersenCommitmentChip        src/chips/pedersen.rs:49-62, 64-134   (over csrc/circuit/ecc.hpp)
//   BoardChip / BoardCircuit      src/chips/board.rs:194-321, 331-500 ; src/circuits/board.rs:14-51
//   ShotChip / ShotCircuit        src/chips/shot.rs:28-51, 179-297, 308-536 ; src/circuits/shot.rs:14-53
// Gate, constraint and region names are the reference's (its MockProver tests assert them: tests/golden/mock_fixtures.json).
#pragma once
#include "ecc.hpp"
#include "game.hpp"

namespace bzc {

// ---- bitify (src/chips/bitify.rs) ---------------------------------------------------------------------------------
struct BitifyConfig {
    Column bits, lc1, e2, fixed;
    Selector selector;
};
inline BitifyConfig bitify_configure(ConstraintSystem& meta, const char* gate_name, Column bits, Column lc1, Column e2, Column fixed) {
    BitifyConfig c{bits, lc1, e2, fixed, meta.selector()};
    meta.create_gate(gate_name, [&](VirtualCells& vc) {
        const Expr one = constant(Fp::one());
        const Expr bit = vc.query_advice(bits, 0);
        const Expr e2_exp = vc.query_advice(e2, 0);
        const Expr e2_next = vc.query_advice(e2, 1);
        const Expr lc1_exp = vc.query_advice(lc1, 0);
        const Expr lc1_next = vc.query_advice(lc1, 1);
        const Expr selector = vc.query_selector(c.selector);
        return with_selector(selector, {{"Constrain bit is boolean", bit * (one - bit)},
                                        {"Start from 1, doubling", e2_exp + e2_exp - e2_next},
                                        {"If bit is 1, e2 added to sum", bit * e2_exp + lc1_exp - lc1_next}});
    });
    return c;
}
inline BitifyConfig num2bits_configure(ConstraintSystem& meta, Column bits, Column lc1, Column e2, Column fixed) {
    return bitify_configure(meta, "num2bits", bits, lc1, e2, fixed);
}
inline BitifyConfig bits2num_configure(ConstraintSystem& meta, Column bits, Column lc1, Column e2, Column fixed) {
    return bitify_configure(meta, "bits2num", bits, lc1, e2, fixed);
}
// Num2BitsChip::synthesize: B + 1 rows; the running sum's last cell is constrained equal to `value`
inline std::vector<AssignedCell> num2bits_synthesize(const BitifyConfig& c, Layouter& layouter, const AssignedCell& value,
                                                     const std::vector<Fp>& bits_in) {
    return layouter.assign_region("num2bits", [&](Region& region) {
        AssignedCell lc1 = region.assign_advice_from_constant(c.lc1, 0, Fp::zero());
        AssignedCell e2 = region.assign_advice_from_constant(c.e2, 0, Fp::one());
        std::vector<AssignedCell> bits;
        for (size_t i = 0; i < bits_in.size(); i++) {
            region.enable_selector(c.selector, i);
            const AssignedCell bit = region.assign_advice(c.bits, i, bits_in[i]);
            bits.push_back(bit);
            const Fp next_lc1 = lc1.value + bit.value * e2.value;
            const Fp next_e2 = e2.value + e2.value;
            lc1 = region.assign_advice(c.lc1, i + 1, next_lc1);
            e2 = region.assign_advice(c.e2, i + 1, next_e2);
        }
        region.constrain_equal(value.cell, lc1.cell);
        return bits;
    });
}
// Bits2NumChip::synthesize: copies the bits in, returns the recomposed cell
inline AssignedCell bits2num_synthesize(const BitifyConfig& c, Layouter& layouter, const std::vector<AssignedCell>& bits_in) {
    return layouter.assign_region("bits2num", [&](Region& region) {
        AssignedCell lc1 = region.assign_advice_from_constant(c.lc1, 0, Fp::zero());
        AssignedCell e2 = region.assign_advice_from_constant(c.e2, 0, Fp::one());
        for (size_t i = 0; i < bits_in.size(); i++) {
            region.enable_selector(c.selector, i);
            const AssignedCell bit = region.copy_advice(bits_in[i], c.bits, i);
            const Fp next_lc1 = lc1.value + bit.value * e2.value;
            const Fp next_e2 = e2.value + e2.value;
            lc1 = region.assign_advice(c.lc1, i + 1, next_lc1);
            e2 = region.assign_advice(c.e2, i + 1, next_e2);
        }
        return lc1;
    });
}

// ---- placement (src/chips/placement.rs) --------------------------------------------------------------------------
struct PlacementConfig {
    int S;
    Column bits, bit_sum, full_window_sum, fixed;
    Selector s_input, s_sum_bits, s_adjacency, s_permute, s_constrain;
};
inline PlacementConfig placement_configure(ConstraintSystem& meta, int S, Column bits, Column bit_sum, Column full_window_sum, Column fixed) {
    PlacementConfig c;
    c.S = S, c.bits = bits, c.bit_sum = bit_sum, c.full_window_sum = full_window_sum, c.fixed = fixed;
    c.s_input = meta.selector();
    c.s_sum_bits = meta.selector();
    c.s_adjacency = meta.selector();
    c.s_permute = meta.selector();
    c.s_constrain = meta.selector();
    meta.create_gate("sum inputted H, V bits", [&](VirtualCells& vc) {
        const Expr horizontal = vc.query_advice(bit_sum, 0);
        const Expr vertical = vc.query_advice(full_window_sum, 0);
        const Expr sum = vc.query_advice(bits, 0);
        const Expr selector = vc.query_selector(c.s_input);
        return with_selector(selector, {{"h + v = sum", sum - (horizontal + vertical)}});
    });
    meta.create_gate("placement bit count", [&](VirtualCells& vc) {
        const Expr bit = vc.query_advice(bits, 0);
        const Expr prev = vc.query_advice(bit_sum, -1);
        const Expr sum = vc.query_advice(bit_sum, 0);
        const Expr selector = vc.query_selector(c.s_sum_bits);
        return with_selector(selector, {{"Running Sum: Bits", bit + prev - sum}});
    });
    meta.create_gate("adjacency bit count", [&](VirtualCells& vc) {
        Expr bit_count = vc.query_advice(bits, 0);
        for (int i = 1; i < S; i++) {
            const Expr bit = vc.query_advice(bits, i);
            bit_count = bit_count + bit;
        }
        const Expr prev_full_window_count = vc.query_advice(full_window_sum, -1);
        const Expr full_window_count = vc.query_advice(full_window_sum, 0);
        auto exp_pow = [](const Expr& base, int pw) {
            Expr e = base;
            if (pw == 0) {
                e = constant(Fp::one());
            } else {
                for (int i = 2; i <= pw; i++) e = e * base;
            }
            return e;
        };
        // degree-S indicator of bit_count == S through Lagrange interpolation over 0..S (src/chips/placement.rs:187-204)
        auto interpolate_incrementor = [&](const Expr& x) {
            std::vector<Fp> points, evals;
            for (int i = 0; i <= S; i++) {
                points.push_back(Fp::from_u64((uint64_t)i));
                evals.push_back(i == S ? Fp::one() : Fp::zero());
            }
            const std::vector<Fp> interpolated = lagrange_interpolate(points, evals);
            Expr value = constant(Fp::zero());
            for (size_t i = 0; i < interpolated.size(); i++) value = value + constant(interpolated[i]) * exp_pow(x, (int)i);
            return value;
        };
        const Expr selector = vc.query_selector(c.s_adjacency);
        const Expr constraint = full_window_count - prev_full_window_count - interpolate_incrementor(bit_count);
        return with_selector(selector, {{"Full Window Running Sum", constraint}});
    });
    meta.create_gate("permute adjaceny bit count", [&](VirtualCells& vc) {
        const Expr previous = vc.query_advice(full_window_sum, -1);
        const Expr current = vc.query_advice(full_window_sum, 0);
        const Expr selector = vc.query_selector(c.s_permute);
        return with_selector(selector, {{"Premute Full Window Running Sum", previous - current}});
    });
    meta.create_gate("running sum constraints", [&](VirtualCells& vc) {
        const Expr ship_len = constant(Fp::from_u64((uint64_t)S));
        const Expr one = constant(Fp::one());
        const Expr bit_count = vc.query_advice(bit_sum, 0);
        const Expr full_window_count = vc.query_advice(full_window_sum, 0);
        const Expr selector = vc.query_selector(c.s_constrain);
        return with_selector(selector, {{"Placed ship of correct length", bit_count - ship_len}, {"One full bit window", full_window_count - one}});
    });
    return c;
}
// compute_placement_trace (src/chips/placement.rs:380-419): [bit_sum, full_window_sum] as small integers
inline void compute_placement_trace(const BinaryValue& ship, int S, uint64_t bit_sum[BOARD_SIZE], uint64_t window[BOARD_SIZE]) {
    uint64_t acc = 0;
    for (int i = 0; i < BOARD_SIZE; i++) {
        acc += ship.bit(i);
        bit_sum[i] = acc;
    }
    auto increment = [&](int off) -> uint64_t {
        int cnt = 0;
        for (int j = off; j < off + S; j++) cnt += ship.bit(j);   // a window never leaves the 256-bit value
        return cnt == S ? 1 : 0;
    };
    window[0] = increment(0);
    for (int i = 1; i < BOARD_SIZE; i++) window[i] = (i % 10 + S > 10) ? window[i - 1] : window[i - 1] + increment(i);
}
inline void placement_synthesize(const PlacementConfig& c, Layouter& layouter, const BinaryValue& ship, const std::vector<AssignedCell>& horizontal,
                                 const std::vector<AssignedCell>& vertical) {
    uint64_t t_bits[BOARD_SIZE], t_win[BOARD_SIZE];
    compute_placement_trace(ship, c.S, t_bits, t_win);
    // load_bits
    const std::vector<AssignedCell> assigned_bits = layouter.assign_region("permute and collapse bit decompositions", [&](Region& region) {
        std::vector<AssignedCell> assigned;
        for (int i = 0; i < BOARD_SIZE; i++) {
            region.enable_selector(c.s_input, i);
            region.copy_advice(horizontal[i], c.bit_sum, i);
            region.copy_advice(vertical[i], c.full_window_sum, i);
            assigned.push_back(region.assign_advice(c.bits, i, ship.bit_fp(i)));
        }
        return assigned;
    });
    // placement_sums
    const std::pair<AssignedCell, AssignedCell> state = layouter.assign_region("placement running sum trace", [&](Region& region) {
        AssignedCell bit_sum = region.assign_advice_from_constant(c.bit_sum, 0, Fp::zero());
        AssignedCell full_window_sum = region.assign_advice_from_constant(c.full_window_sum, 0, Fp::zero());
        for (int i = 0; i < BOARD_SIZE; i++) region.copy_advice(assigned_bits[i], c.bits, i + 1);
        bit_sum = region.assign_advice(c.bit_sum, 1, Fp::from_u64(t_bits[0]));
        full_window_sum = region.assign_advice(c.full_window_sum, 1, Fp::from_u64(t_win[0]));
        region.enable_selector(c.s_sum_bits, 1);
        region.enable_selector(c.s_adjacency, 1);
        for (int offset = 2; offset <= BOARD_SIZE; offset++) {
            const int adjusted = offset - 1;
            bit_sum = region.assign_advice(c.bit_sum, offset, Fp::from_u64(t_bits[adjusted]));
            full_window_sum = region.assign_advice(c.full_window_sum, offset, Fp::from_u64(t_win[adjusted]));
            region.enable_selector(c.s_sum_bits, offset);
            if (adjusted % 10 + c.S > 10) {
                region.enable_selector(c.s_permute, offset);
            } else {
                region.enable_selector(c.s_adjacency, offset);
            }
        }
        return std::make_pair(bit_sum, full_window_sum);
    });
    // assign_constraint
    layouter.assign_region("constrain running sum output", [&](Region& region) {
        region.copy_advice(state.first, c.bit_sum, 0);
        region.copy_advice(state.second, c.full_window_sum, 0);
        region.enable_selector(c.s_constrain, 0);
        return 0;
    });
}

// ---- transpose (src/chips/transpose.rs) --------------------------------------------------------------------------
struct TransposeConfig {
    Column permuted_bits[10];
    Column transposed_bits;
    Selector selector;
};
inline TransposeConfig transpose_configure(ConstraintSystem& meta, const Column permuted_bits[10], Column transposed_bits) {
    TransposeConfig c;
    for (int i = 0; i < 10; i++) c.permuted_bits[i] = permuted_bits[i];
    c.transposed_bits = transposed_bits;
    c.selector = meta.selector();
    meta.create_gate("transpose row constraint", [&](VirtualCells& vc) {
        const Expr zero = constant(Fp::zero());
        const Expr one = constant(Fp::one());
        Expr transposed_bit = zero;
        for (int i = 0; i < 10; i++) transposed_bit = transposed_bit + vc.query_advice(permuted_bits[i], 0);
        const Expr transposed_trace = vc.query_advice(transposed_bits, 0);
        const Expr selector = vc.query_selector(c.selector);
        return with_selector(selector, {{"Constrain trace value integrity", transposed_trace - transposed_bit},
                                        {"Constrain transposition of bit", (one - transposed_bit) * transposed_bit}});
    });
    return c;
}
inline std::vector<AssignedCell> transpose_synthesize(const TransposeConfig& c, Layouter& layouter, const BinaryValue& board,
                                                      const std::vector<std::vector<AssignedCell>>& placements) {
    return layouter.assign_region("Transpose ship commitments", [&](Region& region) {
        for (int col = 0; col < 10; col++) {
            for (int row = 0; row < BOARD_SIZE; row++) {
                const int transposed_index = (col % 2 == 1) ? row % 10 * 10 + row / 10 : row;
                region.copy_advice(placements[col][transposed_index], c.permuted_bits[col], row);
            }
        }
        std::vector<AssignedCell> assigned;
        for (int row = 0; row < BOARD_SIZE; row++) {
            assigned.push_back(region.assign_advice(c.transposed_bits, row, board.bit_fp(row)));
            region.enable_selector(c.selector, row);
        }
        return assigned;
    });
}

// ---- pedersen (src/chips/pedersen.rs) ----------------------------------------------------------------------------
struct PedersenConfig {
    TableColumn table_idx;
    EccConfig ecc;
};
inline PedersenConfig pedersen_configure(ConstraintSystem& meta, const Column advice[10], const Column lagrange[8], TableColumn table_idx) {
    PedersenConfig c;
    c.table_idx = table_idx;
    const LookupRangeCheckConfig range_check = lookup_range_check_configure(meta, advice[9], table_idx);
    c.ecc = ecc_configure(meta, advice, lagrange, range_check);
    return c;
}
struct BoardFixedBases {  // src/utils/constants/fixed_bases.rs:16-87: BoardCommitV (base-field scalar), BoardCommitR (full-width)
    FixedBase v, r;
};
// PedersenCommitmentChip::synthesize: table load, then [value]V + [trapdoor]R
inline EccPoint pedersen_synthesize(const PedersenConfig& c, Layouter& layouter, const BoardFixedBases& bases, const AssignedCell& value,
                                    const Fq& trapdoor) {
    layouter.assign_table("table_idx", [&](Table& table) {
        for (uint64_t index = 0; index < (1u << 10); index++) table.assign_cell(c.table_idx, index, Fp::from_u64(index));
    });
    const EccPoint commitment = mul_fixed_base_field_elem(c.ecc, layouter, value, bases.v);   // [v] BoardCommitV
    const EccPoint blind = mul_fixed_full_width(c.ecc, layouter, trapdoor, bases.r);          // [rcv] BoardCommitR
    return ecc_add(c.ecc, layouter, commitment, blind);                                        // "cv"
}
// native pedersen_commit (src/utils/pedersen.rs:17-28) from the same window tables: [m]V + [t]R, affine
inline Aff pedersen_commit_native(const BoardFixedBases& bases, const Fp& message, const Fq& trapdoor) {
    const auto mc = message.canon();
    if (!Fq::lt_mod(mc.data())) throw GameError("pedersen_commit: message repr is not a canonical scalar");   // from_repr(..).unwrap()
    const auto tc = trapdoor.canon();
    const std::vector<unsigned> mw = decompose_word_3bit(mc.data(), ECC_NUM_WINDOWS), tw = decompose_word_3bit(tc.data(), ECC_NUM_WINDOWS);
    Jac acc = jac_identity();
    for (int w = 0; w < ECC_NUM_WINDOWS; w++) {
        acc = jac_add_mixed(acc, bases.v.points[w][mw[w]]);
        acc = jac_add_mixed(acc, bases.r.points[w][tw[w]]);
    }
    return to_affine(acc);
}

// ---- board (src/chips/board.rs, src/circuits/board.rs) -----------------------------------------------------------
struct BoardConfig {
    BitifyConfig num2bits[10];
    BitifyConfig bits2num;
    PlacementConfig placement[5];
    TransposeConfig transpose;
    PedersenConfig pedersen;
    Column advice[11];
    Column fixed[8];
    TableColumn table_idx;
    Column instance;
    Selector selectors[1];
};
inline BoardConfig board_configure(ConstraintSystem& meta) {
    BoardConfig c;
    for (int i = 0; i < 11; i++) {
        c.advice[i] = meta.advice_column();
        meta.enable_equality(c.advice[i]);
    }
    for (int i = 0; i < 8; i++) c.fixed[i] = meta.fixed_column();
    meta.enable_constant(c.fixed[0]);
    c.table_idx = meta.lookup_table_column();
    c.instance = meta.instance_column();
    meta.enable_equality(c.instance);
    c.selectors[0] = meta.selector();
    for (int i = 0; i < 10; i++) c.num2bits[i] = num2bits_configure(meta, c.advice[0], c.advice[1], c.advice[2], c.fixed[0]);
    c.bits2num = bits2num_configure(meta, c.advice[0], c.advice[1], c.advice[2], c.fixed[0]);
    static const int ship_len[5] = {5, 4, 3, 3, 2};
    for (int i = 0; i < 5; i++) c.placement[i] = placement_configure(meta, ship_len[i], c.advice[0], c.advice[1], c.advice[2], c.fixed[0]);
    c.transpose = transpose_configure(meta, c.advice, c.advice[10]);
    c.pedersen = pedersen_configure(meta, c.advice, c.fixed, c.table_idx);
    meta.create_gate("Commitment orientation H OR V == 0 constraint", [&](VirtualCells& vc) {
        std::vector<Expr> commitments;
        for (int i = 0; i < 10; i++) commitments.push_back(vc.query_advice(c.advice[i], 0));
        const Expr selector = vc.query_selector(c.selectors[0]);
        return with_selector(selector, {{"Aircraft Carrier H OR V == 0", commitments[0] * commitments[1]},
                                        {"Battleship H OR V == 0", commitments[2] * commitments[3]},
                                        {"Cruiser H OR V == 0", commitments[4] * commitments[5]},
                                        {"Submarine H OR V == 0", commitments[6] * commitments[7]},
                                        {"Destroyer H OR V == 0", commitments[8] * commitments[9]}});
    });
    return c;
}
struct BoardInput {  // BoardCircuit::new(ship_commitments, board, trapdoor): src/circuits/board.rs:63-73
    BinaryValue ship_commitments[10];
    BinaryValue board;
    Fq trapdoor;
};
// BoardChip::synthesize (src/chips/board.rs:331-363).  Returns the commitment point the instance column is tied to.
inline Aff board_synthesize(const BoardConfig& c, Layouter& layouter, const BoardFixedBases& bases, const BoardInput& in) {
    BinaryValue ships[5];
    for (int i = 0; i < 5; i++) ships[i] = in.ship_commitments[2 * i].zip(in.ship_commitments[2 * i + 1]);
    // load_commitments
    const std::vector<AssignedCell> assigned_commitments = layouter.assign_region("load ship placements", [&](Region& region) {
        std::vector<AssignedCell> cells;
        for (int i = 0; i < 10; i++) cells.push_back(region.assign_advice(c.advice[i], 0, Fp::from_u128(in.ship_commitments[i].lower_u128())));
        region.enable_selector(c.selectors[0], 0);
        return cells;
    });
    // decompose_commitments
    std::vector<std::vector<AssignedCell>> placements;
    for (int i = 0; i < 10; i++) {
        std::vector<Fp> bits(BOARD_SIZE);
        for (int j = 0; j < BOARD_SIZE; j++) bits[j] = in.ship_commitments[i].bit_fp(j);
        placements.push_back(num2bits_synthesize(c.num2bits[i], layouter, assigned_commitments[i], bits));
    }
    // synth_placements
    for (int i = 0; i < 5; i++) placement_synthesize(c.placement[i], layouter, ships[i], placements[2 * i], placements[2 * i + 1]);
    // transpose_placements, recompose_board, commit_board
    const std::vector<AssignedCell> transposed_bits = transpose_synthesize(c.transpose, layouter, in.board, placements);
    const AssignedCell transposed = bits2num_synthesize(c.bits2num, layouter, transposed_bits);
    const EccPoint commitment = pedersen_synthesize(c.pedersen, layouter, bases, transposed, in.trapdoor);
    layouter.constrain_instance(commitment.x.cell, c.instance, 0);
    layouter.constrain_instance(commitment.y.cell, c.instance, 1);
    return commitment.value();
}

// ---- shot (src/chips/shot.rs, src/circuits/shot.rs) --------------------------------------------------------------
struct ShotConfig {
    BitifyConfig num2bits[2];
    PedersenConfig pedersen;
    Column advice[10];
    Column input;            // the eleventh, unused advice column (src/chips/shot.rs:188-189)
    Column fixed[8];
    TableColumn table_idx;
    Column instance;
    Selector selectors[3];
};
inline ShotConfig shot_configure(ConstraintSystem& meta) {
    ShotConfig c;
    for (int i = 0; i < 10; i++) {
        c.advice[i] = meta.advice_column();
        meta.enable_equality(c.advice[i]);
    }
    c.input = meta.advice_column();
    meta.enable_equality(c.input);
    for (int i = 0; i < 8; i++) c.fixed[i] = meta.fixed_column();
    meta.enable_constant(c.fixed[0]);
    c.table_idx = meta.lookup_table_column();
    c.instance = meta.instance_column();
    meta.enable_equality(c.instance);
    for (int i = 0; i < 3; i++) c.selectors[i] = meta.selector();
    for (int i = 0; i < 2; i++) c.num2bits[i] = num2bits_configure(meta, c.advice[5], c.advice[6], c.advice[7], c.fixed[0]);
    c.pedersen = pedersen_configure(meta, c.advice, c.fixed, c.table_idx);
    meta.create_gate("boolean hit assertion", [&](VirtualCells& vc) {
        const Expr assertion = vc.query_advice(c.advice[4], 0);
        const Expr one = constant(Fp::one());
        const Expr constraint = (one - assertion) * assertion;
        const Expr selector = vc.query_selector(c.selectors[0]);
        return with_selector(selector, {{"asserted hit value is boolean", constraint}});
    });
    meta.create_gate("shot running sum row", [&](VirtualCells& vc) {
        const Expr hit_bit = vc.query_advice(c.advice[5], 0);
        const Expr shot_bit = vc.query_advice(c.advice[6], 0);
        const Expr shot_sum = vc.query_advice(c.advice[7], 0);
        const Expr hit_sum = vc.query_advice(c.advice[8], 0);
        const Expr prev_shot_sum = vc.query_advice(c.advice[7], -1);
        const Expr prev_hit_sum = vc.query_advice(c.advice[8], -1);
        const Expr shot_constraint = shot_bit + prev_shot_sum - shot_sum;
        const Expr hit_constraint = hit_bit * shot_bit + prev_hit_sum - hit_sum;
        const Expr selector = vc.query_selector(c.selectors[1]);
        return with_selector(selector, {{"running sum of flipped bits in shot", shot_constraint}, {"running sum of hits against board", hit_constraint}});
    });
    meta.create_gate("constrain shot running sum output", [&](VirtualCells& vc) {
        const Expr hit_assertion = vc.query_advice(c.advice[5], 0);
        const Expr shot_count = vc.query_advice(c.advice[6], 0);
        const Expr hit_count = vc.query_advice(c.advice[7], 0);
        const Expr shot_constraint = constant(Fp::one()) - shot_count;
        const Expr hit_constraint = hit_assertion - hit_count;
        const Expr selector = vc.query_selector(c.selectors[2]);
        return with_selector(selector, {{"Shot only fires at one board cell", shot_constraint},
                                        {"Public hit assertion matches private witness", hit_constraint}});
    });
    return c;
}
struct ShotInput {  // ShotCircuit::new(board, trapdoor, shot, hit): src/circuits/shot.rs:65-78
    BinaryValue board;
    Fq trapdoor;
    BinaryValue shot, hit;
};
// compute_shot_trace (src/chips/shot.rs:28-51): [shot_trace, hit_trace]
inline void compute_shot_trace(const BinaryValue& board, const BinaryValue& shot, uint64_t shot_trace[BOARD_SIZE], uint64_t hit_trace[BOARD_SIZE]) {
    uint64_t s = 0, h = 0;
    for (int i = 0; i < BOARD_SIZE; i++) {
        s += shot.bit(i);
        h += (board.bit(i) && shot.bit(i)) ? 1 : 0;
        shot_trace[i] = s;
        hit_trace[i] = h;
    }
}
// ShotChip::synthesize (src/chips/shot.rs:308-354)
inline Aff shot_synthesize(const ShotConfig& c, Layouter& layouter, const BoardFixedBases& bases, const ShotInput& in) {
    const Fp board_state = Fp::from_u128(in.board.lower_u128());
    const Aff board_commitment = pedersen_commit_native(bases, board_state, in.trapdoor);
    const Fp shot_commitment = Fp::from_u128(in.shot.lower_u128());
    uint64_t shot_trace[BOARD_SIZE], hit_trace[BOARD_SIZE];
    compute_shot_trace(in.board, in.shot, shot_trace, hit_trace);
    // load_advice
    const std::vector<AssignedCell> inputs = layouter.assign_region("load private ShotChip advice values", [&](Region& region) {
        std::vector<AssignedCell> v;
        v.push_back(region.assign_advice(c.advice[4], 0, board_state));
        v.push_back(region.assign_advice(c.advice[4], 1, board_commitment.x));
        v.push_back(region.assign_advice(c.advice[4], 2, board_commitment.y));
        v.push_back(region.assign_advice(c.advice[4], 3, shot_commitment));
        v.push_back(region.assign_advice(c.advice[4], 4, Fp::from_u128(in.hit.lower_u128())));
        region.enable_selector(c.selectors[0], 4);
        return v;
    });
    // decompose
    std::vector<Fp> bbits(BOARD_SIZE), sbits(BOARD_SIZE);
    for (int j = 0; j < BOARD_SIZE; j++) bbits[j] = in.board.bit_fp(j), sbits[j] = in.shot.bit_fp(j);
    const std::vector<AssignedCell> board_bits = num2bits_synthesize(c.num2bits[0], layouter, inputs[0], bbits);
    const std::vector<AssignedCell> shot_bits = num2bits_synthesize(c.num2bits[1], layouter, inputs[3], sbits);
    // running_sums
    const std::pair<AssignedCell, AssignedCell> sums = layouter.assign_region("shot running sum", [&](Region& region) {
        AssignedCell shot_sum = region.assign_advice_from_constant(c.advice[7], 0, Fp::zero());
        AssignedCell hit_sum = region.assign_advice_from_constant(c.advice[8], 0, Fp::zero());
        for (int i = 0; i < BOARD_SIZE; i++) {
            region.copy_advice(board_bits[i], c.advice[5], i + 1);
            region.copy_advice(shot_bits[i], c.advice[6], i + 1);
            shot_sum = region.assign_advice(c.advice[7], i + 1, Fp::from_u64(shot_trace[i]));
            hit_sum = region.assign_advice(c.advice[8], i + 1, Fp::from_u64(hit_trace[i]));
            region.enable_selector(c.selectors[1], i + 1);
        }
        return std::make_pair(shot_sum, hit_sum);
    });
    // running_sum_output
    layouter.assign_region("shot running sum output checks", [&](Region& region) {
        region.copy_advice(inputs[4], c.advice[5], 0);
        region.copy_advice(sums.first, c.advice[6], 0);
        region.copy_advice(sums.second, c.advice[7], 0);
        region.enable_selector(c.selectors[2], 0);
        return 0;
    });
    // commit_board
    const EccPoint commitment = pedersen_synthesize(c.pedersen, layouter, bases, inputs[0], in.trapdoor);
    layouter.constrain_instance(commitment.x.cell, c.instance, 0);
    layouter.constrain_instance(commitment.y.cell, c.instance, 1);
    layouter.constrain_instance(inputs[3].cell, c.instance, 2);
    layouter.constrain_instance(inputs[4].cell, c.instance, 3);
    return commitment.value();
}

}  // namespace bzc
