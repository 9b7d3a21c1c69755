// The ECC chip and lookup range check the reference wires into both circuits through
// `PedersenCommitmentChip` (src/chips/pedersen.rs:49-62 configure, :64-134 synthesize): halo2_gadgets 0.2.0
// `ecc::chip::EccChip<BoardFixedBases>` + `utilities::lookup_range_check::LookupRangeCheckConfig<_, 10>`
// (UPSTREAM, un-vendored: Cargo.lock:363-366).  Its 19 gates (Shot gates 2-20, Board gates 37-55:
// src/circuits/shot.rs:303, src/circuits/board.rs:271,682) are restated here from the halo2 book's ECC chapter
// (witness point, incomplete / complete addition, variable-base mul, fixed-base mul) and from memory of the crate:
// PARITY with upstream's exact polynomials / query order is UNPINNED (nothing of the crate is on disk).  What the
// reference does pin, and tests check: 19 gates, the 8 pedersen regions with "complete point addition" last and its
// result x in advice 2 at offset 1 (src/circuits/shot.rs:684, src/circuits/board.rs:867), the fixed-base tables
// (U / Z rows of tests/golden/fixed_bases.json, src/utils/constants/fixed_bases/board_commit_{v,r}.rs).
//
// Only what the reference's circuits execute gets witness code: base-field-element fixed-base mul ([v]V), full-width
// fixed-base mul ([r]R), complete addition, the 10-bit lookup range check.  The variable-base and short fixed-base
// gates are configured (they shape the constraint system) but never enabled, exactly as in the reference.
#pragma once
#include <atomic>
#include <thread>

#include "layouter.hpp"

namespace bzc {

static constexpr int ECC_H = 8;                 // 2^FIXED_BASE_WINDOW_SIZE
static constexpr int ECC_WINDOW_BITS = 3;       // FIXED_BASE_WINDOW_SIZE
static constexpr int ECC_NUM_WINDOWS = 85;      // src/utils/constants.rs:4
static constexpr int LOOKUP_K = 10;             // src/utils/constants.rs:10

// ---- fixed-base tables (halo2_gadgets ecc::chip::constants: compute_lagrange_coeffs, find_zs_and_us) -------------
struct FixedBase {
    Aff generator;
    std::vector<std::array<Aff, ECC_H>> points;     // window w, k: [(k+2) 8^w]B, last window [k 8^84 - sum_{j<84} 2 8^j]B
    std::vector<std::array<Fp, ECC_H>> lagrange;    // x-coordinate interpolation coefficients per window, low to high
    std::vector<uint64_t> z;                        // smallest z with y_k + z square and z - y_k non-square for all k
    std::vector<std::array<Fp, ECC_H>> u;           // u^2 = y_k + z
};

inline void fixed_base_window_points(const Aff& base, int num_windows, std::vector<std::array<Aff, ECC_H>>& out) {
    std::vector<Jac> all;
    Jac pw = to_jac(base);               // [8^w]B
    Jac sum_lower = jac_identity();      // sum_{j<w} [8^j]B
    for (int w = 0; w < num_windows; w++) {
        if (w < num_windows - 1) {
            Jac m = jac_double(pw);      // [2 8^w]B
            for (int k = 0; k < ECC_H; k++) {
                all.push_back(m);
                m = jac_add(m, pw);
            }
            sum_lower = jac_add(sum_lower, pw);
        } else {
            const Jac off = jac_neg(jac_double(sum_lower));   // -[sum 2 8^j]B
            Jac m = off;                                       // k = 0
            for (int k = 0; k < ECC_H; k++) {
                all.push_back(m);
                m = jac_add(m, pw);
            }
        }
        pw = jac_double(jac_double(jac_double(pw)));
    }
    const std::vector<Aff> aff = batch_normalize(all);
    out.resize(num_windows);
    for (int w = 0; w < num_windows; w++) {
        for (int k = 0; k < ECC_H; k++) out[w][k] = aff[(size_t)w * ECC_H + k];
    }
}

inline FixedBase make_fixed_base(const Aff& base, int num_windows = ECC_NUM_WINDOWS, unsigned threads = 0) {
    FixedBase fb;
    fb.generator = base;
    fixed_base_window_points(base, num_windows, fb.points);
    std::vector<Fp> xs_pts;
    for (int k = 0; k < ECC_H; k++) xs_pts.push_back(Fp::from_u64((uint64_t)k));
    fb.lagrange.resize(num_windows);
    fb.z.assign(num_windows, 0);
    fb.u.resize(num_windows);
    auto work = [&](int w) {
        std::vector<Fp> xs(ECC_H);
        for (int k = 0; k < ECC_H; k++) xs[k] = fb.points[w][k].x;
        const std::vector<Fp> co = lagrange_interpolate(xs_pts, xs);
        for (int k = 0; k < ECC_H; k++) fb.lagrange[w][k] = co[k];
        for (uint64_t z = 0;; z++) {
            const Fp zf = Fp::from_u64(z);
            bool ok = true;
            for (int k = 0; k < ECC_H && ok; k++) {
                const Fp y = fb.points[w][k].y;
                const int a = (y + zf).jacobi(), b = (zf - y).jacobi();
                ok = (a >= 0) && (b < 0);      // y + z a square (0 included), z - y not a square
            }
            if (ok) {
                fb.z[w] = z;
                for (int k = 0; k < ECC_H; k++) {
                    if (!(fb.points[w][k].y + zf).sqrt(&fb.u[w][k])) throw std::logic_error("fixed-base u: not a square");
                }
                break;
            }
        }
    };
    if (!threads) threads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (unsigned t = 0; t < threads; t++) {
        pool.emplace_back([&] {
            for (int w; (w = next.fetch_add(1)) < num_windows;) work(w);
        });
    }
    for (auto& t : pool) t.join();
    return fb;
}

// ---- configs --------------------------------------------------------------------------------------------------
struct LookupRangeCheckConfig {
    Selector q_lookup, q_running, q_bitshift;
    Column running_sum;
    TableColumn table_idx;
};
struct AddIncompleteConfig {
    Selector q_add_incomplete;
    Column x_p, y_p, x_qr, y_qr;
};
struct AddConfig {
    Selector q_add;
    Column x_p, y_p, x_qr, y_qr, lambda, alpha, beta, gamma, delta;
};
struct MulFixedConfig {
    Selector q_running_sum;            // RunningSumConfig::q_range_check
    Column lagrange_coeffs[ECC_H];
    Column fixed_z;
    Column window, u;
    AddConfig add;
    AddIncompleteConfig add_incomplete;
};
struct EccConfig {
    Column advices[10];
    Selector q_point, q_point_non_id;
    AddIncompleteConfig add_incomplete;
    AddConfig add;
    Selector q_mul_hi[3], q_mul_lo[3], q_mul_decompose_var, q_mul_overflow, q_mul_lsb;
    MulFixedConfig mul_fixed;
    Selector q_mul_fixed_full, q_mul_fixed_short, q_mul_fixed_base_field;
    Column canon_advices[3];
    LookupRangeCheckConfig lookup;
};

static const uint64_t T_P_LIMBS[2] = {0x992d30ed00000001ull, 0x224698fc094cf91bull};   // p - 2^254
static const uint64_t T_Q_LIMBS[2] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull};   // q - 2^254
inline Fp fp_two_pow(unsigned e) {
    Fp r = Fp::one();
    for (unsigned i = 0; i < e; i++) r = r.dbl();
    return r;
}
inline Fp fp_from_limbs2(const uint64_t* l) { return Fp::from_raw_reduce({l[0], l[1], 0, 0}); }

// LookupRangeCheckConfig::configure (halo2_gadgets utilities/lookup_range_check.rs; call site src/chips/pedersen.rs:56-57)
inline LookupRangeCheckConfig lookup_range_check_configure(ConstraintSystem& meta, Column running_sum, TableColumn table_idx) {
    meta.enable_equality(running_sum);
    LookupRangeCheckConfig cfg;
    cfg.q_lookup = meta.complex_selector();
    cfg.q_running = meta.complex_selector();
    cfg.q_bitshift = meta.selector();
    cfg.running_sum = running_sum;
    cfg.table_idx = table_idx;
    meta.lookup([&](VirtualCells& vc) {
        const Expr q_lookup = vc.query_selector(cfg.q_lookup);
        const Expr q_running = vc.query_selector(cfg.q_running);
        const Expr z_cur = vc.query_advice(running_sum, 0);
        // running-sum word: z_i - 2^K z_{i+1}
        const Expr z_next = vc.query_advice(running_sum, 1);
        const Expr running_sum_word = z_cur - z_next * Fp::from_u64(1ull << LOOKUP_K);
        const Expr running_sum_lookup = q_running * running_sum_word;
        // short range check: the word is witnessed directly
        const Expr q_short = constant(Fp::one()) - q_running;
        const Expr short_lookup = q_short * z_cur;
        std::vector<std::pair<Expr, TableColumn>> out;
        out.push_back({q_lookup * (running_sum_lookup + short_lookup), table_idx});
        return out;
    });
    meta.create_gate("Short lookup bitshift", [&](VirtualCells& vc) {
        const Expr q_bitshift = vc.query_selector(cfg.q_bitshift);
        const Expr word = vc.query_advice(running_sum, -1);
        const Expr shifted_word = vc.query_advice(running_sum, 0);
        const Expr inv_two_pow_s = vc.query_advice(running_sum, 1);
        const Fp two_pow_k = Fp::from_u64(1ull << LOOKUP_K);
        return with_selector(q_bitshift, {{"", word * two_pow_k * inv_two_pow_s - shifted_word}});
    });
    return cfg;
}

namespace ecc_detail {
// DoubleAndAdd helpers of the variable-base incomplete rows
struct DoubleAndAdd {
    Column x_a, x_p, lambda_1, lambda_2;
    Expr x_r(VirtualCells& vc, int rot) const {
        const Expr x_a_ = vc.query_advice(x_a, rot);
        const Expr x_p_ = vc.query_advice(x_p, rot);
        const Expr l1 = vc.query_advice(lambda_1, rot);
        return square(l1) - x_a_ - x_p_;
    }
    Expr Y_A(VirtualCells& vc, int rot) const {
        const Expr x_a_ = vc.query_advice(x_a, rot);
        const Expr l1 = vc.query_advice(lambda_1, rot);
        const Expr l2 = vc.query_advice(lambda_2, rot);
        return (l1 + l2) * (x_a_ - x_r(vc, rot));
    }
};
inline Expr ternary(const Expr& a, const Expr& b, const Expr& c) {
    const Expr one_minus_a = constant(Fp::one()) - a;
    return a * b + one_minus_a * c;
}
inline void mul_incomplete_configure(ConstraintSystem& meta, Column z, Column x_a, Column x_p, Column y_p, Column lambda1, Column lambda2,
                                     Selector out_q[3]) {
    meta.enable_equality(z);
    meta.enable_equality(lambda1);
    const Selector q1 = meta.selector(), q2 = meta.selector(), q3 = meta.selector();
    out_q[0] = q1, out_q[1] = q2, out_q[2] = q3;
    const DoubleAndAdd daa{x_a, x_p, lambda1, lambda2};
    const Fp two_inv = Fp::from_u64(2).inv();
    auto y_a = [&](VirtualCells& vc, int rot) { return daa.Y_A(vc, rot) * two_inv; };
    auto for_loop = [&](VirtualCells& vc, const Expr& y_a_next) -> Constraints {
        const Expr one = constant(Fp::one());
        const Expr z_cur = vc.query_advice(z, 0);
        const Expr z_prev = vc.query_advice(z, -1);
        const Expr x_a_cur = vc.query_advice(x_a, 0);
        const Expr x_a_next = vc.query_advice(x_a, 1);
        const Expr x_p_cur = vc.query_advice(x_p, 0);
        const Expr y_p_cur = vc.query_advice(y_p, 0);
        const Expr lambda1_cur = vc.query_advice(lambda1, 0);
        const Expr lambda2_cur = vc.query_advice(lambda2, 0);
        const Expr y_a_cur = y_a(vc, 0);
        const Expr k = z_cur - z_prev * Fp::from_u64(2);
        const Expr bc = bool_check(k);
        const Expr gradient_1 = lambda1_cur * (x_a_cur - x_p_cur) - y_a_cur + (k * Fp::from_u64(2) - one) * y_p_cur;
        const Expr secant_line = square(lambda2_cur) - x_a_next - daa.x_r(vc, 0) - x_a_cur;
        const Expr gradient_2 = lambda2_cur * (x_a_cur - x_a_next) - y_a_cur - y_a_next;
        return Constraints{{"bool_check", bc}, {"gradient_1", gradient_1}, {"secant_line", secant_line}, {"gradient_2", gradient_2}};
    };
    meta.create_gate("q_mul_1 == 1 checks", [&](VirtualCells& vc) {
        const Expr q = vc.query_selector(q1);
        const Expr y_a_next = y_a(vc, 1);
        const Expr y_a_witnessed = vc.query_advice(lambda1, 0);
        return with_selector(q, {{"init y_a", y_a_witnessed - y_a_next}});
    });
    meta.create_gate("q_mul_2 == 1 checks", [&](VirtualCells& vc) {
        const Expr q = vc.query_selector(q2);
        const Expr y_a_next = y_a(vc, 1);
        const Expr x_p_cur = vc.query_advice(x_p, 0);
        const Expr x_p_next = vc.query_advice(x_p, 1);
        const Expr y_p_cur = vc.query_advice(y_p, 0);
        const Expr y_p_next = vc.query_advice(y_p, 1);
        Constraints c{{"x_p_check", x_p_cur - x_p_next}, {"y_p_check", y_p_cur - y_p_next}};
        for (auto& e : for_loop(vc, y_a_next)) c.push_back(e);
        return with_selector(q, c);
    });
    meta.create_gate("q_mul_3 == 1 checks", [&](VirtualCells& vc) {
        const Expr q = vc.query_selector(q3);
        const Expr y_a_final = vc.query_advice(lambda1, 1);
        return with_selector(q, for_loop(vc, y_a_final));
    });
}
}  // namespace ecc_detail

// mul_fixed::Config::coords_check
inline Constraints mul_fixed_coords_check(const MulFixedConfig& c, VirtualCells& vc, const Expr& window) {
    const Expr y_p = vc.query_advice(c.add.y_p, 0);
    const Expr x_p = vc.query_advice(c.add.x_p, 0);
    const Expr z = vc.query_fixed(c.fixed_z, 0);
    const Expr u = vc.query_advice(c.u, 0);
    std::vector<Expr> window_pow;
    for (int pw = 0; pw < ECC_H; pw++) {
        Expr acc = constant(Fp::one());
        for (int i = 0; i < pw; i++) acc = acc * window;
        window_pow.push_back(acc);
    }
    Expr interpolated_x = constant(Fp::zero());
    for (int i = 0; i < ECC_H; i++) interpolated_x = interpolated_x + window_pow[i] * vc.query_fixed(c.lagrange_coeffs[i], 0);
    const Expr x_check = interpolated_x - x_p;
    const Expr y_check = square(u) - y_p - z;
    const Expr on_curve = square(y_p) - square(x_p) * x_p - constant(Fp::from_u64(5));
    return Constraints{{"check x", x_check}, {"check y", y_check}, {"on-curve", on_curve}};
}

// EccChip::configure (halo2_gadgets ecc/chip.rs; call site src/chips/pedersen.rs:58-59)
inline EccConfig ecc_configure(ConstraintSystem& meta, const Column advices[10], const Column lagrange[ECC_H], const LookupRangeCheckConfig& lookup_cfg) {
    using namespace ecc_detail;
    EccConfig cfg;
    for (int i = 0; i < 10; i++) cfg.advices[i] = advices[i];
    cfg.lookup = lookup_cfg;
    const Expr b5 = constant(Fp::from_u64(5));

    // witness_point::Config::configure(meta, advices[0], advices[1])
    {
        const Column x = advices[0], y = advices[1];
        cfg.q_point = meta.selector();
        cfg.q_point_non_id = meta.selector();
        auto curve_eqn = [&](VirtualCells& vc, Expr& xo, Expr& yo) {
            xo = vc.query_advice(x, 0);
            yo = vc.query_advice(y, 0);
            return square(yo) - (square(xo) * xo) - b5;
        };
        meta.create_gate("witness point", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(cfg.q_point);
            Expr xe, ye;
            const Expr eq = curve_eqn(vc, xe, ye);
            return Constraints{{"x == 0 v on_curve", q * xe * eq}, {"y == 0 v on_curve", q * ye * eq}};
        });
        meta.create_gate("witness non-identity point", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(cfg.q_point_non_id);
            Expr xe, ye;
            const Expr eq = curve_eqn(vc, xe, ye);
            return with_selector(q, {{"on_curve", eq}});
        });
    }
    // add_incomplete::Config::configure(meta, advices[0..4])
    {
        AddIncompleteConfig& c = cfg.add_incomplete;
        c.x_p = advices[0], c.y_p = advices[1], c.x_qr = advices[2], c.y_qr = advices[3];
        meta.enable_equality(c.x_p);
        meta.enable_equality(c.y_p);
        meta.enable_equality(c.x_qr);
        meta.enable_equality(c.y_qr);
        c.q_add_incomplete = meta.selector();
        meta.create_gate("incomplete addition", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(c.q_add_incomplete);
            const Expr x_p = vc.query_advice(c.x_p, 0);
            const Expr y_p = vc.query_advice(c.y_p, 0);
            const Expr x_q = vc.query_advice(c.x_qr, 0);
            const Expr y_q = vc.query_advice(c.y_qr, 0);
            const Expr x_r = vc.query_advice(c.x_qr, 1);
            const Expr y_r = vc.query_advice(c.y_qr, 1);
            // (x_r + x_q + x_p)(x_p - x_q)^2 - (y_p - y_q)^2 = 0
            const Expr poly1 = (x_r + x_q + x_p) * (x_p - x_q) * (x_p - x_q) - square(y_p - y_q);
            // (y_r + y_q)(x_p - x_q) - (y_p - y_q)(x_q - x_r) = 0
            const Expr poly2 = (y_r + y_q) * (x_p - x_q) - (y_p - y_q) * (x_q - x_r);
            return with_selector(q, {{"x_r", poly1}, {"y_r", poly2}});
        });
    }
    // add::Config::configure(meta, advices[0..9])
    {
        AddConfig& c = cfg.add;
        c.x_p = advices[0], c.y_p = advices[1], c.x_qr = advices[2], c.y_qr = advices[3], c.lambda = advices[4];
        c.alpha = advices[5], c.beta = advices[6], c.gamma = advices[7], c.delta = advices[8];
        meta.enable_equality(c.x_p);
        meta.enable_equality(c.y_p);
        meta.enable_equality(c.x_qr);
        meta.enable_equality(c.y_qr);
        c.q_add = meta.selector();
        meta.create_gate("complete addition", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(c.q_add);
            const Expr x_p = vc.query_advice(c.x_p, 0);
            const Expr y_p = vc.query_advice(c.y_p, 0);
            const Expr x_q = vc.query_advice(c.x_qr, 0);
            const Expr y_q = vc.query_advice(c.y_qr, 0);
            const Expr x_r = vc.query_advice(c.x_qr, 1);
            const Expr y_r = vc.query_advice(c.y_qr, 1);
            const Expr lambda = vc.query_advice(c.lambda, 0);
            const Expr alpha = vc.query_advice(c.alpha, 0);   // inv0(x_q - x_p)
            const Expr beta = vc.query_advice(c.beta, 0);     // inv0(x_p)
            const Expr gamma = vc.query_advice(c.gamma, 0);   // inv0(x_q)
            const Expr delta = vc.query_advice(c.delta, 0);   // inv0(y_p + y_q) if x_q = x_p
            const Expr if_alpha = (x_q - x_p) * alpha;
            const Expr if_beta = x_p * beta;
            const Expr if_gamma = x_q * gamma;
            const Expr if_delta = (y_q + y_p) * delta;
            const Expr one = constant(Fp::one()), two = constant_u64(2), three = constant_u64(3);
            const Expr x_q_minus_x_p = x_q - x_p;
            const Expr poly1 = x_q_minus_x_p * (x_q_minus_x_p * lambda - (y_q - y_p));
            const Expr tangent_line = (two * y_p) * lambda - three * square(x_p);
            const Expr poly2 = (one - if_alpha) * tangent_line;
            const Expr secant_line = square(lambda) - x_p - x_q - x_r;
            const Expr poly3a = x_p * x_q * (x_q - x_p) * secant_line;
            const Expr line_y = lambda * (x_p - x_r) - y_p - y_r;
            const Expr poly3b = x_p * x_q * (x_q - x_p) * line_y;
            const Expr poly3c = x_p * x_q * (y_q + y_p) * secant_line;
            const Expr poly3d = x_p * x_q * (y_q + y_p) * line_y;
            const Expr poly4a = (one - if_beta) * (x_r - x_q);
            const Expr poly4b = (one - if_beta) * (y_r - y_q);
            const Expr poly5a = (one - if_gamma) * (x_r - x_p);
            const Expr poly5b = (one - if_gamma) * (y_r - y_p);
            const Expr poly6a = (one - if_alpha - if_delta) * x_r;
            const Expr poly6b = (one - if_alpha - if_delta) * y_r;
            return with_selector(q, {{"1", poly1}, {"2", poly2}, {"3a", poly3a}, {"3b", poly3b}, {"3c", poly3c}, {"3d", poly3d},
                                     {"4a", poly4a}, {"4b", poly4b}, {"5a", poly5a}, {"5b", poly5b}, {"6a", poly6a}, {"6b", poly6b}});
        });
    }
    // mul::Config::configure(meta, add, range_check, advices): variable-base scalar mul (configured, never enabled here)
    {
        mul_incomplete_configure(meta, advices[9], advices[3], advices[0], advices[1], advices[4], advices[5], cfg.q_mul_hi);
        mul_incomplete_configure(meta, advices[6], advices[7], advices[0], advices[1], advices[8], advices[2], cfg.q_mul_lo);
        // complete::Config::configure(meta, advices[9], add)
        const Column z_complete = advices[9];
        meta.enable_equality(z_complete);
        cfg.q_mul_decompose_var = meta.selector();
        meta.create_gate("Decompose scalar for complete bits of variable-base mul", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(cfg.q_mul_decompose_var);
            const Expr z_prev = vc.query_advice(z_complete, -1);
            const Expr z_next = vc.query_advice(z_complete, 1);
            const Expr k = z_next - constant_u64(2) * z_prev;
            const Expr bc = bool_check(k);
            const Expr base_y = vc.query_advice(z_complete, 0);
            const Expr y_p = vc.query_advice(cfg.add.y_p, -1);
            const Expr y_switch = ternary(k, base_y - y_p, base_y + y_p);
            return with_selector(q, {{"bool_check", bc}, {"y_switch", y_switch}});
        });
        // overflow::Config::configure(meta, range_check, advices[6..9])
        const Column ov[3] = {advices[6], advices[7], advices[8]};
        for (int i = 0; i < 3; i++) meta.enable_equality(ov[i]);
        cfg.q_mul_overflow = meta.selector();
        meta.create_gate("overflow checks", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(cfg.q_mul_overflow);
            const Expr one = constant(Fp::one());
            const Expr two_pow_124 = constant(fp_two_pow(124));
            const Expr two_pow_130 = two_pow_124 * constant(fp_two_pow(6));
            const Expr z_0 = vc.query_advice(ov[0], -1);
            const Expr z_130 = vc.query_advice(ov[0], 0);
            const Expr eta = vc.query_advice(ov[0], 1);
            const Expr k_254 = vc.query_advice(ov[1], -1);
            const Expr alpha = vc.query_advice(ov[1], 0);
            const Expr s_minus_lo_130 = vc.query_advice(ov[1], 1);
            const Expr s = vc.query_advice(ov[2], 0);
            const Expr s_check = s - (alpha + k_254 * two_pow_130);
            const Expr t_q = constant(fp_from_limbs2(T_Q_LIMBS));
            const Expr recovery = z_0 - alpha - t_q;
            const Expr lo_zero = k_254 * (z_130 - two_pow_124);
            const Expr s_minus_lo_130_check = k_254 * s_minus_lo_130;
            const Expr canonicity = (one - k_254) * (one - z_130 * eta) * s_minus_lo_130;
            return with_selector(q, {{"s_check", s_check}, {"recovery", recovery}, {"lo_zero", lo_zero},
                                     {"s_minus_lo_130_check", s_minus_lo_130_check}, {"canonicity", canonicity}});
        });
        cfg.q_mul_lsb = meta.selector();
        meta.create_gate("LSB check", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(cfg.q_mul_lsb);
            const Expr z_1 = vc.query_advice(z_complete, 0);
            const Expr z_0 = vc.query_advice(z_complete, 1);
            const Expr x_p = vc.query_advice(cfg.add.x_p, 0);
            const Expr y_p = vc.query_advice(cfg.add.y_p, 0);
            const Expr base_x = vc.query_advice(cfg.add.x_p, 1);
            const Expr base_y = vc.query_advice(cfg.add.y_p, 1);
            const Expr lsb = z_0 - z_1 * Fp::from_u64(2);
            const Expr one_minus_lsb = constant(Fp::one()) - lsb;
            const Expr bc = bool_check(lsb);
            const Expr lsb_x = (lsb * x_p) + one_minus_lsb * (x_p - base_x);
            const Expr lsb_y = (lsb * y_p) + one_minus_lsb * (y_p + base_y);
            return with_selector(q, {{"bool_check", bc}, {"lsb_x", lsb_x}, {"lsb_y", lsb_y}});
        });
    }
    // mul_fixed::Config::configure(meta, lagrange_coeffs, advices[4], advices[5], add, add_incomplete)
    {
        MulFixedConfig& c = cfg.mul_fixed;
        c.window = advices[4];
        c.u = advices[5];
        c.add = cfg.add;
        c.add_incomplete = cfg.add_incomplete;
        for (int i = 0; i < ECC_H; i++) c.lagrange_coeffs[i] = lagrange[i];
        meta.enable_equality(c.window);
        meta.enable_equality(c.u);
        c.q_running_sum = meta.selector();
        // RunningSumConfig::configure(meta, q_running_sum, window)
        meta.enable_equality(c.window);
        meta.create_gate("range check", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(c.q_running_sum);
            const Expr z_cur = vc.query_advice(c.window, 0);
            const Expr z_next = vc.query_advice(c.window, 1);
            const Expr word = z_cur - z_next * Fp::from_u64(1ull << ECC_WINDOW_BITS);
            return with_selector(q, {{"", range_check(word, 1ull << ECC_WINDOW_BITS)}});
        });
        c.fixed_z = meta.fixed_column();
        meta.create_gate("Running sum coordinates check", [&](VirtualCells& vc) {
            const Expr q = vc.query_selector(c.q_running_sum);
            const Expr z_cur = vc.query_advice(c.window, 0);
            const Expr z_next = vc.query_advice(c.window, 1);
            const Expr word = z_cur - z_next * Fp::from_u64((uint64_t)ECC_H);
            return with_selector(q, mul_fixed_coords_check(c, vc, word));
        });
    }
    // mul_fixed::full_width::Config::configure
    cfg.q_mul_fixed_full = meta.selector();
    meta.create_gate("Full-width fixed-base scalar mul", [&](VirtualCells& vc) {
        const Expr q = vc.query_selector(cfg.q_mul_fixed_full);
        const Expr window = vc.query_advice(cfg.mul_fixed.window, 0);
        Constraints c = mul_fixed_coords_check(cfg.mul_fixed, vc, window);
        c.push_back({"window range check", range_check(window, (uint64_t)ECC_H)});
        return with_selector(q, c);
    });
    // mul_fixed::short::Config::configure
    cfg.q_mul_fixed_short = meta.selector();
    meta.create_gate("Short fixed-base mul gate", [&](VirtualCells& vc) {
        const Expr q = vc.query_selector(cfg.q_mul_fixed_short);
        const Expr y_p = vc.query_advice(cfg.add.y_p, 0);
        const Expr y_a = vc.query_advice(cfg.add.y_qr, 0);
        const Expr last_window = vc.query_advice(cfg.mul_fixed.u, 0);
        const Expr sign = vc.query_advice(cfg.mul_fixed.window, 0);
        const Expr one = constant(Fp::one());
        const Expr last_window_check = bool_check(last_window);
        const Expr sign_check = square(sign) - one;
        const Expr y_check = (y_p - y_a) * (y_p + y_a);
        const Expr negation_check = sign * y_p - y_a;
        return with_selector(q, {{"last_window_check", last_window_check}, {"sign_check", sign_check}, {"y_check", y_check},
                                 {"negation_check", negation_check}});
    });
    // mul_fixed::base_field_elem::Config::configure(meta, advices[6..9], range_check, mul_fixed)
    for (int i = 0; i < 3; i++) {
        cfg.canon_advices[i] = advices[6 + i];
        meta.enable_equality(cfg.canon_advices[i]);
    }
    cfg.q_mul_fixed_base_field = meta.selector();
    meta.create_gate("Canonicity checks", [&](VirtualCells& vc) {
        const Expr q = vc.query_selector(cfg.q_mul_fixed_base_field);
        const Expr alpha = vc.query_advice(cfg.canon_advices[0], -1);
        const Expr z_84_alpha = vc.query_advice(cfg.canon_advices[2], -1);
        const Expr alpha_0 = alpha - (z_84_alpha * fp_two_pow(252));
        const Expr alpha_1 = vc.query_advice(cfg.canon_advices[1], 0);
        const Expr alpha_2 = vc.query_advice(cfg.canon_advices[2], 0);
        const Expr alpha_0_prime = vc.query_advice(cfg.canon_advices[0], 0);
        const Expr z_13_alpha_0_prime = vc.query_advice(cfg.canon_advices[0], 1);
        const Expr z_44_alpha = vc.query_advice(cfg.canon_advices[1], 1);
        const Expr z_43_alpha = vc.query_advice(cfg.canon_advices[2], 1);
        // decomposition checks
        const Expr alpha_1_range_check = range_check(alpha_1, 1 << 2);
        const Expr alpha_2_range_check = bool_check(alpha_2);
        const Expr z_84_alpha_check = z_84_alpha - (alpha_1 + alpha_2 * Fp::from_u64(1 << 2));
        // alpha_0_prime = alpha_0 + 2^130 - t_p
        const Expr two_pow_130 = constant(fp_two_pow(130));
        const Expr t_p = constant(fp_from_limbs2(T_P_LIMBS));
        const Expr alpha_0_prime_check = alpha_0_prime - (alpha_0 + two_pow_130 - t_p);
        // canonicity: MSB = 1 => alpha_1 = 0, alpha_0 < t_p
        const Expr alpha_0_hi_120 = z_44_alpha - z_84_alpha * constant(fp_two_pow(120));
        const Expr a_43 = z_43_alpha - z_44_alpha * Fp::from_u64(8);
        Constraints c{{"MSB = 1 => alpha_1 = 0", alpha_2 * alpha_1},
                      {"MSB = 1 => alpha_0_hi_120 = 0", alpha_2 * alpha_0_hi_120},
                      {"MSB = 1 => a_43 = 0 or 1", alpha_2 * bool_check(a_43)},
                      {"MSB = 1 => z_13_alpha_0_prime = 0", alpha_2 * z_13_alpha_0_prime},
                      {"alpha_1_range_check", alpha_1_range_check},
                      {"alpha_2_range_check", alpha_2_range_check},
                      {"z_84_alpha_check", z_84_alpha_check},
                      {"alpha_0_prime check", alpha_0_prime_check}};
        return with_selector(q, c);
    });
    return cfg;
}

// ---- witness --------------------------------------------------------------------------------------------------
struct EccPoint {
    AssignedCell x, y;
    Aff value() const { return Aff{x.value, y.value}; }
};

inline Fp inv0(const Fp& v) { return v.is_zero() ? Fp::zero() : v.inv(); }

// add::Config::assign_region: complete addition P + Q, result in (x_qr, y_qr) at offset + 1
inline EccPoint add_assign_region(const AddConfig& c, const EccPoint& p, const EccPoint& q, size_t offset, Region& region) {
    region.enable_selector(c.q_add, offset);
    region.copy_advice(p.x, c.x_p, offset);
    region.copy_advice(p.y, c.y_p, offset);
    region.copy_advice(q.x, c.x_qr, offset);
    region.copy_advice(q.y, c.y_qr, offset);
    const Fp x_p = p.x.value, y_p = p.y.value, x_q = q.x.value, y_q = q.y.value;
    std::vector<Fp> inv{x_q - x_p, x_p, x_q, y_q + y_p};
    batch_invert(inv);
    const Fp alpha = inv[0], beta = inv[1], gamma = inv[2];
    const Fp delta = (x_q == x_p) ? inv[3] : Fp::zero();
    region.assign_advice(c.alpha, offset, alpha);
    region.assign_advice(c.beta, offset, beta);
    region.assign_advice(c.gamma, offset, gamma);
    region.assign_advice(c.delta, offset, delta);
    Fp lambda;
    if (x_q != x_p) {
        lambda = (y_q - y_p) * alpha;
    } else if (!y_p.is_zero()) {
        lambda = x_p.sqr() * Fp::from_u64(3) * (y_p.dbl()).inv();
    } else {
        lambda = Fp::zero();
    }
    region.assign_advice(c.lambda, offset, lambda);
    Fp x_r, y_r;
    if (x_p.is_zero()) {
        x_r = x_q, y_r = y_q;
    } else if (x_q.is_zero()) {
        x_r = x_p, y_r = y_p;
    } else if (x_q == x_p && y_q == -y_p) {
        x_r = Fp::zero(), y_r = Fp::zero();
    } else {
        x_r = lambda.sqr() - x_p - x_q;
        y_r = lambda * (x_p - x_r) - y_p;
    }
    EccPoint r;
    r.x = region.assign_advice(c.x_qr, offset + 1, x_r);
    r.y = region.assign_advice(c.y_qr, offset + 1, y_r);
    return r;
}

// add_incomplete::Config::assign_region: P + Q for distinct non-identity points with different x
// (`known_sum`: P + Q already computed by the caller -- the gate has no slope cell, only the three points)
inline EccPoint add_incomplete_assign_region(const AddIncompleteConfig& c, const EccPoint& p, const EccPoint& q, size_t offset, Region& region,
                                             const Aff* known_sum = nullptr) {
    region.enable_selector(c.q_add_incomplete, offset);
    const Fp x_p = p.x.value, y_p = p.y.value, x_q = q.x.value, y_q = q.y.value;
    if (!region.shape_pass && ((x_p.is_zero() && y_p.is_zero()) || (x_q.is_zero() && y_q.is_zero()) || x_p == x_q))
        throw SynthesisError("incomplete addition: exceptional case");
    region.copy_advice(p.x, c.x_p, offset);
    region.copy_advice(p.y, c.y_p, offset);
    region.copy_advice(q.x, c.x_qr, offset);
    region.copy_advice(q.y, c.y_qr, offset);
    Fp x_r, y_r;
    if (known_sum) {
        x_r = known_sum->x, y_r = known_sum->y;
    } else {
        const Fp lambda = (y_q - y_p) * (x_q - x_p).inv();
        x_r = lambda.sqr() - x_p - x_q;
        y_r = lambda * (x_p - x_r) - y_p;
    }
    EccPoint r;
    r.x = region.assign_advice(c.x_qr, offset + 1, x_r);
    r.y = region.assign_advice(c.y_qr, offset + 1, y_r);
    return r;
}

// mul_fixed::Config::assign_region_inner: fixed constants, the per-window points [(k_w + 2) 8^w]B, their running
// incomplete sum; returns (acc over windows 0..83, the most significant window's point)
inline std::pair<EccPoint, EccPoint> mul_fixed_assign_region_inner(const MulFixedConfig& c, Region& region, size_t offset,
                                                                    const std::vector<unsigned>& windows, const FixedBase& base,
                                                                    Selector coords_check_toggle) {
    const int NW = (int)base.points.size();
    if ((int)windows.size() != NW) throw SynthesisError("fixed-base mul: window count");
    // assign_fixed_constants
    for (int w = 0; w < NW; w++) {
        region.enable_selector(coords_check_toggle, offset + w);
        for (int k = 0; k < ECC_H; k++) region.assign_fixed(c.lagrange_coeffs[k], offset + w, base.lagrange[w][k]);
        region.assign_fixed(c.fixed_z, offset + w, Fp::from_u64(base.z[w]));
    }
    auto process_window = [&](int w) {
        const unsigned k = windows[w];
        const Aff& m = base.points[w][k];
        if (m.x.is_zero()) throw SynthesisError("fixed-base mul: window point with x = 0");
        EccPoint pt;
        pt.x = region.assign_advice(c.add.x_p, offset + w, m.x);
        pt.y = region.assign_advice(c.add.y_p, offset + w, m.y);
        region.assign_advice(c.u, offset + w, base.u[w][k]);
        return pt;
    };
    // The running sums acc_w = sum_{j <= w} m_j are what the incomplete-addition rows witness (no slope cell): they are
    // accumulated in Jacobian coordinates and normalised with ONE field inversion, instead of one inversion per row.
    std::vector<Jac> sums(NW - 1);
    sums[0] = to_jac(base.points[0][windows[0]]);
    for (int w = 1; w < NW - 1; w++) sums[w] = jac_add_mixed(sums[w - 1], base.points[w][windows[w]]);
    const std::vector<Aff> acc_aff = region.shape_pass ? std::vector<Aff>(NW - 1, Aff{Fp::zero(), Fp::zero()}) : batch_normalize(sums);
    EccPoint acc = process_window(0);                       // initialize_accumulator
    for (int w = 1; w < NW - 1; w++) {                      // add_incomplete over the lower windows
        const EccPoint mul_b = process_window(w);
        acc = add_incomplete_assign_region(c.add_incomplete, mul_b, acc, offset + w, region, &acc_aff[w]);
    }
    const EccPoint mul_b = process_window(NW - 1);          // process_msb
    return {acc, mul_b};
}

inline std::vector<unsigned> decompose_word_3bit(const uint64_t* canon, int num_windows) {
    std::vector<unsigned> out(num_windows);
    for (int w = 0; w < num_windows; w++) {
        unsigned v = 0;
        for (int b = 0; b < ECC_WINDOW_BITS; b++) {
            const int bit = w * ECC_WINDOW_BITS + b;
            if (bit < 256) v |= (unsigned)((canon[bit / 64] >> (bit % 64)) & 1) << b;
        }
        out[w] = v;
    }
    return out;
}
inline Fp fp_from_shifted(const uint64_t* canon, unsigned shift) {   // canon >> shift as a field element
    uint64_t r[4] = {0, 0, 0, 0};
    const unsigned ws = shift / 64, bs = shift % 64;
    for (unsigned i = 0; i + ws < 4; i++) {
        r[i] = canon[i + ws] >> bs;
        if (bs && i + ws + 1 < 4) r[i] |= canon[i + ws + 1] << (64 - bs);
    }
    return Fp::from_raw_reduce({r[0], r[1], r[2], r[3]});
}

// LookupRangeCheckConfig::witness_check(value, num_words, strict): region "Witness element"
inline std::vector<AssignedCell> lookup_witness_check(const LookupRangeCheckConfig& c, Layouter& layouter, const Fp& value, int num_words, bool strict) {
    return layouter.assign_region("Witness element", [&](Region& region) {
        AssignedCell z = region.assign_advice(c.running_sum, 0, value);
        std::vector<AssignedCell> zs{z};
        const auto bits = value.canon();
        const Fp inv_two_pow_k = Fp::from_u64(1ull << LOOKUP_K).inv();
        for (int idx = 0; idx < num_words; idx++) {
            region.enable_selector(c.q_lookup, idx);
            region.enable_selector(c.q_running, idx);
            uint64_t word = 0;
            for (int b = 0; b < LOOKUP_K; b++) {
                const int bit = idx * LOOKUP_K + b;
                word |= ((bits[bit / 64] >> (bit % 64)) & 1) << b;
            }
            const Fp z_next = (z.value - Fp::from_u64(word)) * inv_two_pow_k;
            z = region.assign_advice(c.running_sum, idx + 1, z_next);
            zs.push_back(z);
        }
        if (strict) region.constrain_constant(zs.back().cell, Fp::zero());
        return zs;
    });
}

// mul_fixed::base_field_elem::Config::assign: [alpha]B for a base-field element alpha already in a cell
inline EccPoint mul_fixed_base_field_elem(const EccConfig& cfg, Layouter& layouter, const AssignedCell& scalar, const FixedBase& base) {
    const MulFixedConfig& mf = cfg.mul_fixed;
    const int NW = ECC_NUM_WINDOWS;
    std::vector<AssignedCell> running_sum;
    auto r1 = layouter.assign_region("Base-field elem fixed-base mul (incomplete addition)", [&](Region& region) {
        const size_t offset = 0;
        // RunningSumConfig::copy_decompose(alpha, strict = true, 255 bits, 85 windows)
        AssignedCell z = region.copy_advice(scalar, mf.window, offset);
        running_sum.assign(1, z);
        for (int idx = 0; idx < NW; idx++) region.enable_selector(mf.q_running_sum, offset + idx);
        const auto canon = z.value.canon();
        const std::vector<unsigned> words = decompose_word_3bit(canon.data(), NW);
        const Fp two_pow_k_inv = Fp::from_u64(1ull << ECC_WINDOW_BITS).inv();
        for (int i = 0; i < NW; i++) {
            const Fp z_next = (z.value - Fp::from_u64(words[i])) * two_pow_k_inv;
            z = region.assign_advice(mf.window, offset + i + 1, z_next);
            running_sum.push_back(z);
        }
        region.constrain_constant(running_sum.back().cell, Fp::zero());
        return mul_fixed_assign_region_inner(mf, region, offset, words, base, mf.q_running_sum);
    });
    const EccPoint result = layouter.assign_region("Base-field elem fixed-base mul (complete addition)", [&](Region& region) {
        return add_assign_region(mf.add, r1.second, r1.first, 0, region);
    });
    // canonicity of alpha
    const AssignedCell& alpha = running_sum[0];
    const AssignedCell &z_43 = running_sum[43], &z_44 = running_sum[44], &z_84 = running_sum[84];
    const Fp alpha_0 = alpha.value - z_84.value * fp_two_pow(252);
    const Fp alpha_0_prime_v = alpha_0 + fp_two_pow(130) - fp_from_limbs2(T_P_LIMBS);
    const std::vector<AssignedCell> zs = lookup_witness_check(cfg.lookup, layouter, alpha_0_prime_v, 13, false);
    const AssignedCell &alpha_0_prime = zs[0], &z_13_alpha_0_prime = zs[13];
    layouter.assign_region("Canonicity checks", [&](Region& region) {
        region.enable_selector(cfg.q_mul_fixed_base_field, 1);
        region.copy_advice(alpha, cfg.canon_advices[0], 0);
        region.copy_advice(z_84, cfg.canon_advices[2], 0);
        region.copy_advice(alpha_0_prime, cfg.canon_advices[0], 1);
        const auto canon = alpha.value.canon();
        const uint64_t alpha_1 = (canon[3] >> 60) & 3, alpha_2 = (canon[3] >> 62) & 1;     // bits 252..253, bit 254
        region.assign_advice(cfg.canon_advices[1], 1, Fp::from_u64(alpha_1));
        region.assign_advice(cfg.canon_advices[2], 1, Fp::from_u64(alpha_2));
        region.copy_advice(z_13_alpha_0_prime, cfg.canon_advices[0], 2);
        region.copy_advice(z_44, cfg.canon_advices[1], 2);
        region.copy_advice(z_43, cfg.canon_advices[2], 2);
        return 0;
    });
    return result;
}

// mul_fixed::full_width::Config::assign: [s]B for a full-width scalar s (Pallas scalar field), witnessed as 85 windows
inline EccPoint mul_fixed_full_width(const EccConfig& cfg, Layouter& layouter, const Fq& scalar, const FixedBase& base) {
    const MulFixedConfig& mf = cfg.mul_fixed;
    const int NW = ECC_NUM_WINDOWS;
    auto r1 = layouter.assign_region("Full-width fixed-base mul (incomplete addition)", [&](Region& region) {
        const size_t offset = 0;
        for (int idx = 0; idx < NW; idx++) region.enable_selector(cfg.q_mul_fixed_full, offset + idx);
        const auto canon = scalar.canon();
        const std::vector<unsigned> windows = decompose_word_3bit(canon.data(), NW);
        for (int idx = 0; idx < NW; idx++) region.assign_advice(mf.window, offset + idx, Fp::from_u64(windows[idx]));
        return mul_fixed_assign_region_inner(mf, region, offset, windows, base, cfg.q_mul_fixed_full);
    });
    return layouter.assign_region("Full-width fixed-base mul (last window, complete addition)", [&](Region& region) {
        return add_assign_region(mf.add, r1.second, r1.first, 0, region);
    });
}

// EccInstructions::add: region "complete point addition"
inline EccPoint ecc_add(const EccConfig& cfg, Layouter& layouter, const EccPoint& a, const EccPoint& b) {
    return layouter.assign_region("complete point addition", [&](Region& region) { return add_assign_region(cfg.add, a, b, 0, region); });
}

}  // namespace bzc
