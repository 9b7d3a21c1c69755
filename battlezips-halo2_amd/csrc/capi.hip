// C ABI entry points (include/bzh2.h): context, base tables, host<->device
// staging around the MSM / NTT drivers, and the host-side helpers.
#include <cstring>
#include <algorithm>
#include <thread>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "curve.cuh"

namespace bzh {
void ntt_cache_drop(bzh_ctx* ctx);
}

using namespace bzh;

static bool valid_curve(int c) { return c >= 0 && c <= 2; }
static bool valid_field(int f) { return f >= 0 && f <= 3; }
static bool valid_form(int f) { return f == BZH_FORM_CANONICAL || f == BZH_FORM_MONTGOMERY; }
static bool valid_mem(int m) { return m == BZH_MEM_HOST || m == BZH_MEM_DEVICE; }
static unsigned field_two_adicity(int f) { return f == BZH_FIELD_BN254_FR ? 28u : (f == BZH_FIELD_BN254_FQ ? 1u : 32u); }

// ---- host helpers (CPU build of the same field templates) ------------------
namespace {
// Staging for BZH_MEM_HOST calls: carve device buffers out of workspace slot 3, copy in, and
// (canonical form) convert to Montgomery on the device.
struct Stager {
    bzh_ctx* ctx;
    int field, form;
    char* cur = nullptr;
    int begin(size_t total_bytes) {
        void* p = nullptr;
        int rc = ws_ensure(ctx, 3, total_bytes + 256, &p);
        cur = (char*)p;
        return rc;
    }
    uint32_t* carve(size_t bytes) {
        uint32_t* p = (uint32_t*)cur;
        cur += (bytes + 63) & ~(size_t)63;
        return p;
    }
    int in(const void* host, size_t elems, uint32_t** dev) {
        *dev = carve(elems * 32);
        if (!elems) return BZH_OK;
        int rc = h2d_small(ctx, *dev, host, elems * 32);
        if (rc) return rc;
        if (form == BZH_FORM_CANONICAL) return field_convert(ctx, field, *dev, elems, 1);
        return BZH_OK;
    }
    int out(void* host, uint32_t* dev, size_t elems) {
        if (!elems) return BZH_OK;
        if (form == BZH_FORM_CANONICAL) {
            int rc = field_convert(ctx, field, dev, elems, 0);
            if (rc) return rc;
        }
        int rc = d2h_async(ctx, host, dev, elems * 32);
        if (rc) return rc;
        return d2h_finish(ctx);
    }
};

template <class P>
static Fe<P> load_host(const uint64_t* p, int form) {
    Fe<P> v;
    for (int i = 0; i < 4; i++) {
        v.l[2 * i] = (uint32_t)p[i];
        v.l[2 * i + 1] = (uint32_t)(p[i] >> 32);
    }
    return form == BZH_FORM_CANONICAL ? fe_to_mont(v) : v;
}
template <class P>
static void store_host(uint64_t* p, Fe<P> v, int form) {
    if (form == BZH_FORM_CANONICAL) v = fe_from_mont(v);
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32);
}

template <class P>
static void jac_to_aff_host(const uint64_t* xyz, size_t n, int form, uint64_t* out) {
    // batch inversion of Z (Montgomery's trick), identity -> (0,0)
    std::vector<Fe<P>> z(n), pref(n);
    Fe<P> run = fe_one<P>();
    for (size_t i = 0; i < n; i++) {
        z[i] = load_host<P>(xyz + 12 * i + 8, form);
        pref[i] = run;
        if (!fe_is_zero(z[i])) run = fe_mul(run, z[i]);
    }
    Fe<P> inv = fe_inv(run);
    for (size_t i = n; i-- > 0;) {
        if (fe_is_zero(z[i])) {
            memset(out + 8 * i, 0, 64);
            continue;
        }
        Fe<P> zi = fe_mul(inv, pref[i]);
        inv = fe_mul(inv, z[i]);
        Fe<P> zi2 = fe_sqr(zi), zi3 = fe_mul(zi2, zi);
        store_host<P>(out + 8 * i, fe_mul(load_host<P>(xyz + 12 * i, form), zi2), form);
        store_host<P>(out + 8 * i + 4, fe_mul(load_host<P>(xyz + 12 * i + 4, form), zi3), form);
    }
}

template <class P>
static void compress_host(const uint64_t* xy, size_t n, int form, uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        uint64_t x[4], y[4];
        store_host<P>(x, load_host<P>(xy + 8 * i, form), BZH_FORM_CANONICAL);
        store_host<P>(y, load_host<P>(xy + 8 * i + 4, form), BZH_FORM_CANONICAL);
        memcpy(out + 32 * i, x, 32);
        out[32 * i + 31] |= (uint8_t)((y[0] & 1) << 7);
    }
}

// sum of n Jacobian points on the host: the combine step of an MSM whose points are split over several GPUs
// (each rank's partial result is one 96-byte point; SURVEY.md section 8e "8-GPU single MSM")
template <class P>
static void jac_sum_host(const uint64_t* xyz, size_t n, int form, uint64_t* out) {
    std::vector<uint64_t> aff(8 * (n ? n : 1));
    jac_to_aff_host<P>(xyz, n, form, aff.data());
    Xyzz<P> acc = xyzz_identity<P>();
    for (size_t i = 0; i < n; i++) {
        Affine<P> a;
        a.x = load_host<P>(aff.data() + 8 * i, form);
        a.y = load_host<P>(aff.data() + 8 * i + 4, form);
        if (aff_is_id(a)) continue;
        xyzz_madd(acc, a);
    }
    Fe<P> X, Y, Z;
    xyzz_to_jacobian(acc, X, Y, Z);
    store_host<P>(out, X, form);
    store_host<P>(out + 4, Y, form);
    store_host<P>(out + 8, Z, form);
}

template <class P>
static void omega_host(unsigned S, uint32_t gen, unsigned log_n, int form, uint64_t* out) {
    // ROOT_OF_UNITY = gen^((p-1) >> S); omega = ROOT^(2^(S - log_n))
    uint32_t e[8];
    uint64_t br = 1;
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)P::mod(i) - br;
        e[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    uint32_t sh[8];
    for (int i = 0; i < 8; i++) {
        uint64_t lo = (S < 32) ? ((uint64_t)e[i] >> S) : 0;
        unsigned src = i + S / 32;
        uint64_t v = 0;
        if (S % 32 == 0) {
            v = src < 8 ? e[src] : 0;
        } else {
            uint64_t a = src < 8 ? e[src] : 0, b = src + 1 < 8 ? e[src + 1] : 0;
            v = ((a | (b << 32)) >> (S % 32)) & 0xffffffffu;
        }
        (void)lo;
        sh[i] = (uint32_t)v;
    }
    Fe<P> g = fe_from_u32<P>(gen);
    Fe<P> root = fe_pow(g, sh);
    for (unsigned i = log_n; i < S; i++) root = fe_sqr(root);
    store_host<P>(out, root, form);
}

template <class PP>
static int permute_pair_host(const uint64_t* input, const uint64_t* table, size_t usable, int form, uint64_t* out_input,
                             uint64_t* out_table) {
    // canonical values as sortable keys (pasta_curves orders field elements by canonical value)
    struct Key {
        uint64_t l[4];
        bool operator<(const Key& o) const {
            for (int i = 3; i >= 0; i--)
                if (l[i] != o.l[i]) return l[i] < o.l[i];
            return false;
        }
        bool operator==(const Key& o) const { return !memcmp(l, o.l, 32); }
    };
    std::vector<Key> a(usable), t(usable);
    if (form == BZH_FORM_CANONICAL) {  // canonical limbs ARE the sort keys: no field arithmetic on the host
        memcpy(a.data(), input, usable * 32);
        memcpy(t.data(), table, usable * 32);
    } else {
        for (size_t i = 0; i < usable; i++) {
            store_host<PP>(a[i].l, load_host<PP>(input + 4 * i, form), BZH_FORM_CANONICAL);
            store_host<PP>(t[i].l, load_host<PP>(table + 4 * i, form), BZH_FORM_CANONICAL);
        }
    }
    // values below 2^64 (range tables, compressed small columns -- the reference's lookups): sort the low limbs as integers
    auto is_small = [](const std::vector<Key>& v) {
        for (const Key& k : v)
            if (k.l[1] | k.l[2] | k.l[3]) return false;
        return true;
    };
    const bool small_a = is_small(a), small_t = is_small(t);
    auto sort_keys = [](std::vector<Key>& v, bool small) {
        if (!small) {
            std::sort(v.begin(), v.end());
            return;
        }
        uint64_t mx = 0;
        for (const Key& k : v) mx = std::max(mx, k.l[0]);
        if (mx < 65536 && v.size() >= 256) {   // a histogram does it (10-bit range table: 1 024 distinct values)
            std::vector<uint32_t> cnt(mx + 1, 0);
            for (const Key& k : v) cnt[k.l[0]]++;
            size_t at = 0;
            for (uint64_t val = 0; val <= mx; val++)
                for (uint32_t c = cnt[val]; c; c--) v[at++].l[0] = val;
            return;
        }
        std::vector<uint64_t> lo(v.size());
        for (size_t i = 0; i < v.size(); i++) lo[i] = v[i].l[0];
        std::sort(lo.begin(), lo.end());
        for (size_t i = 0; i < v.size(); i++) v[i].l[0] = lo[i];
    };
    if (usable >= 4096 && !(small_a && small_t)) {  // the two sorts are independent: a second thread takes the table
        std::thread other([&]() { sort_keys(t, small_t); });
        sort_keys(a, small_a);
        other.join();
    } else {   // (integer / histogram sorts take less than starting a thread does)
        sort_keys(a, small_a);
        sort_keys(t, small_t);
    }
    // leftover multiset = table minus one copy of every distinct input value
    std::vector<Key> s(usable);
    std::vector<size_t> repeated;
    std::vector<char> used(usable, 0);
    size_t tp = 0;
    for (size_t i = 0; i < usable; i++) {
        if (i == 0 || !(a[i] == a[i - 1])) {
            while (tp < usable && t[tp] < a[i]) tp++;
            if (tp >= usable || !(t[tp] == a[i])) return BZH_E_RANGE;  // input value not in the table
            used[tp++] = 1;
            s[i] = a[i];
        } else {
            repeated.push_back(i);
        }
    }
    for (size_t j = 0; j < usable; j++) {  // ascending leftovers go to the repeated rows from the last one backwards
        if (used[j]) continue;
        if (repeated.empty()) return BZH_E_RANGE;
        s[repeated.back()] = t[j];
        repeated.pop_back();
    }
    if (form == BZH_FORM_CANONICAL) {
        memcpy(out_input, a.data(), usable * 32);
        memcpy(out_table, s.data(), usable * 32);
        return BZH_OK;
    }
    for (size_t i = 0; i < usable; i++) {
        store_host<PP>(out_input + 4 * i, load_host<PP>(a[i].l, BZH_FORM_CANONICAL), form);
        store_host<PP>(out_table + 4 * i, load_host<PP>(s[i].l, BZH_FORM_CANONICAL), form);
    }
    return BZH_OK;
}

}  // namespace

extern "C" {

const char* bzh_version(void) { return "bzh2 0.1 (gfx950)"; }

const char* bzh_strerror(int status) {
    switch (status) {
        case BZH_OK: return "ok";
        case BZH_E_ARG: return "invalid argument";
        case BZH_E_OOM: return "out of memory";
        case BZH_E_HIP: return "HIP runtime error";
        case BZH_E_RANGE: return "value out of range";
        case BZH_E_NOGPU: return "no usable GPU";
        case BZH_E_VERIFY: return "proof does not verify";
    }
    return "unknown status";
}

int bzh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ctx_create_common(int device, void* stream, bool own, bzh_ctx** out) {
    if (!out) return BZH_E_ARG;
    *out = nullptr;
    int n = bzh_device_count();
    if (n <= 0) return BZH_E_NOGPU;
    if (device < 0 || device >= n) return BZH_E_ARG;
    bzh_ctx* ctx = new (std::nothrow) bzh_ctx();
    if (!ctx) return BZH_E_OOM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return BZH_E_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return BZH_E_HIP;
        }
        ctx->own_stream = true;
    } else {
        ctx->stream = (hipStream_t)stream;
    }
    *out = ctx;
    return BZH_OK;
}

int bzh_ctx_create(int device, bzh_ctx** out) { return ctx_create_common(device, nullptr, true, out); }
int bzh_ctx_create_on_stream(int device, void* hip_stream, bzh_ctx** out) {
    return ctx_create_common(device, hip_stream, false, out);
}

int bzh_ctx_destroy(bzh_ctx* ctx) {
    if (!ctx) return BZH_E_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ntt_cache_drop(ctx);
    for (int i = 0; i < bzh_ctx::kWsSlots; i++)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    if (ctx->d_add_counter) (void)hipFree(ctx->d_add_counter);
    if (ctx->ped_tbl) (void)hipFree(ctx->ped_tbl);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->pin_big) (void)hipHostFree(ctx->pin_big);
    for (auto& s : ctx->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return BZH_OK;
}

int bzh_ctx_sync(bzh_ctx* ctx) {
    if (!ctx) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BZH_OK;
}

const char* bzh_last_error(const bzh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

static int drain_spans(bzh_ctx* ctx) {
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& s : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            ctx->acc_ms[s.cls] += ms;
            ctx->acc_n[s.cls] += 1;
        }
        ctx->event_pool.push_back(s.a);
        ctx->event_pool.push_back(s.b);
    }
    ctx->spans.clear();
    return BZH_OK;
}

int bzh_ctx_profile(bzh_ctx* ctx, int enable) {
    if (!ctx) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = drain_spans(ctx);
    if (rc) return rc;
    for (int i = 0; i < BZH_T_COUNT; i++) {
        ctx->acc_ms[i] = 0;
        ctx->acc_n[i] = 0;
        ctx->alg_bytes[i] = 0;
    }
    if (enable && !ctx->d_add_counter) {
        BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
        BZH_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_add_counter, 8));
    }
    if (ctx->d_add_counter) BZH_HIP_TRY(ctx, hipMemsetAsync(ctx->d_add_counter, 0, 8, ctx->stream));
    ctx->profiling = enable != 0;
    return BZH_OK;
}

int bzh_ctx_msm_additions(bzh_ctx* ctx, uint64_t* additions) {
    if (!ctx || !additions) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    *additions = 0;
    if (!ctx->d_add_counter) return BZH_OK;
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long v = 0;
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    BZH_HIP_TRY(ctx, hipMemcpy(&v, ctx->d_add_counter, 8, hipMemcpyDeviceToHost));
    *additions = v;
    return BZH_OK;
}

int bzh_ctx_work(bzh_ctx* ctx, double* algorithmic_bytes) {
    if (!ctx || !algorithmic_bytes) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 0; i < BZH_T_COUNT; i++) algorithmic_bytes[i] = ctx->alg_bytes[i];
    return BZH_OK;
}

int bzh_ctx_timings(bzh_ctx* ctx, double* ms, uint64_t* launches) {
    if (!ctx || !ms || !launches) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = drain_spans(ctx);
    if (rc) return rc;
    for (int i = 0; i < BZH_T_COUNT; i++) {
        ms[i] = ctx->acc_ms[i];
        launches[i] = ctx->acc_n[i];
    }
    return BZH_OK;
}

int bzh_bases_upload(bzh_ctx* ctx, int curve, const uint64_t* xy, size_t n, int form, int mem, bzh_bases** out) {
    if (!ctx || !out || (!xy && n) || !valid_curve(curve) || !valid_form(form) || !valid_mem(mem)) return BZH_E_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    bzh_bases* b = new (std::nothrow) bzh_bases();
    if (!b) return BZH_E_OOM;
    b->curve = curve;
    b->n = n;
    b->device = ctx->device;
    hipError_t e = hipMalloc((void**)&b->d_xy, (n ? n : 1) * 64);
    if (e != hipSuccess) {
        delete b;
        ctx->last_error = std::string("hipMalloc(bases): ") + hipGetErrorString(e);
        return BZH_E_OOM;
    }
    e = hipMemcpyAsync(b->d_xy, xy, n * 64, mem == BZH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                       ctx->stream);
    int rc = BZH_OK;
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMemcpyAsync(bases): ") + hipGetErrorString(e);
        rc = BZH_E_HIP;
    }
    if (!rc && form == BZH_FORM_CANONICAL) rc = bases_to_montgomery(ctx, curve, b->d_xy, n);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BZH_E_HIP;
    if (rc) {
        (void)hipFree(b->d_xy);
        delete b;
        return rc;
    }
    *out = b;
    return BZH_OK;
}

int bzh_bases_free(bzh_ctx* ctx, bzh_bases* bases) {
    if (!ctx || !bases) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(bases->d_xy);
    if (bases->d_xy29) (void)hipFree(bases->d_xy29);
    delete bases;
    return BZH_OK;
}

size_t bzh_bases_len(const bzh_bases* bases) { return bases ? bases->n : 0; }

int bzh_bases_precompute(bzh_ctx* ctx, bzh_bases* bases, int window_bits) {
    if (!ctx || !bases || window_bits < 0) return BZH_E_ARG;
    if (bases->device != ctx->device) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bases_precompute(ctx, bases, window_bits);
}

int bzh_msm(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* scalars, size_t n, size_t batch, int form, int mem,
            uint64_t* out_xyz) {
    if (!ctx || !bases || !out_xyz || (!scalars && n && batch) || !valid_form(form) || !valid_mem(mem)) return BZH_E_ARG;
    if (n > bases->n) return BZH_E_ARG;  // halo2: assert_eq!(coeffs.len(), bases.len())
    if (bases->device != ctx->device) return BZH_E_ARG;
    if (batch == 0) return BZH_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (mem == BZH_MEM_DEVICE) return msm_run(ctx, bases, (const uint32_t*)scalars, n, batch, form, (uint32_t*)out_xyz);
    void* stage = nullptr;
    const size_t sbytes = batch * n * 32, obytes = batch * 96;
    int rc = ws_ensure(ctx, 3, sbytes + obytes + 64, &stage);
    if (rc) return rc;
    uint32_t* d_s = (uint32_t*)stage;
    uint32_t* d_o = (uint32_t*)((char*)stage + ((sbytes + 63) & ~(size_t)63));
    if (sbytes) BZH_HIP_TRY(ctx, hipMemcpyAsync(d_s, scalars, sbytes, hipMemcpyHostToDevice, ctx->stream));
    rc = msm_run(ctx, bases, d_s, n, batch, form, d_o);
    if (rc) return rc;
    BZH_HIP_TRY(ctx, hipMemcpyAsync(out_xyz, d_o, obytes, hipMemcpyDeviceToHost, ctx->stream));
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BZH_OK;
}

int bzh_ntt(bzh_ctx* ctx, int field, uint64_t* data, unsigned log_n, size_t batch, const uint64_t* omega,
            const uint64_t* coset_shift, int inverse, int form, int mem) {
    if (!ctx || !data || !omega || !valid_field(field) || !valid_form(form) || !valid_mem(mem)) return BZH_E_ARG;
    if (log_n > field_two_adicity(field) || log_n > 30) return BZH_E_RANGE;
    if (batch == 0) return BZH_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (mem == BZH_MEM_DEVICE) return ntt_run(ctx, field, (uint32_t*)data, log_n, batch, omega, coset_shift, inverse, form);
    void* stage = nullptr;
    const size_t bytes = (batch << log_n) * 32;
    int rc = ws_ensure(ctx, 3, bytes, &stage);
    if (rc) return rc;
    BZH_HIP_TRY(ctx, hipMemcpyAsync(stage, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = ntt_run(ctx, field, (uint32_t*)stage, log_n, batch, omega, coset_shift, inverse, form);
    if (rc) return rc;
    BZH_HIP_TRY(ctx, hipMemcpyAsync(data, stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BZH_OK;
}

// 4 host limbs, canonical -> Montgomery in place
static int host_to_montgomery(int field, uint64_t* v) {
    switch (field) {
        case BZH_FIELD_FP: store_host<FpParams>(v, load_host<FpParams>(v, BZH_FORM_CANONICAL), BZH_FORM_MONTGOMERY); return BZH_OK;
        case BZH_FIELD_FQ: store_host<FqParams>(v, load_host<FqParams>(v, BZH_FORM_CANONICAL), BZH_FORM_MONTGOMERY); return BZH_OK;
        case BZH_FIELD_BN254_FR: store_host<BnFrParams>(v, load_host<BnFrParams>(v, BZH_FORM_CANONICAL), BZH_FORM_MONTGOMERY); return BZH_OK;
        case BZH_FIELD_BN254_FQ: store_host<BnFqParams>(v, load_host<BnFqParams>(v, BZH_FORM_CANONICAL), BZH_FORM_MONTGOMERY); return BZH_OK;
    }
    return BZH_E_ARG;
}

int bzh_coeff_to_extended(bzh_ctx* ctx, int field, const uint64_t* coeffs, unsigned log_n, uint64_t* out, unsigned log_ext,
                          size_t batch, const uint64_t* omega_ext, const uint64_t* coset_shift, int form, int mem) {
    if (!ctx || !coeffs || !out || !omega_ext || !valid_field(field) || !valid_form(form) || !valid_mem(mem)) return BZH_E_ARG;
    if (log_ext > field_two_adicity(field) || log_ext > 30 || log_n > log_ext) return BZH_E_RANGE;
    if (batch == 0) return BZH_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t in_elems = batch << log_n, out_elems = batch << log_ext;
    const uint32_t* d_in = (const uint32_t*)coeffs;
    uint32_t* d_out = (uint32_t*)out;
    void* stage = nullptr;
    if (mem == BZH_MEM_HOST || form == BZH_FORM_CANONICAL) {
        // staged copy of the coefficients (host memory, or canonical device input that may not be written to)
        int rc = ws_ensure(ctx, 3, (in_elems + (mem == BZH_MEM_HOST ? out_elems : 0)) * 32, &stage);
        if (rc) return rc;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(stage, coeffs, in_elems * 32, mem == BZH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                                        ctx->stream));
        d_in = (const uint32_t*)stage;
        if (mem == BZH_MEM_HOST) d_out = (uint32_t*)stage + in_elems * 8;
        if (form == BZH_FORM_CANONICAL) {
            rc = field_convert(ctx, field, (uint32_t*)stage, in_elems, 1);
            if (rc) return rc;
        }
    }
    uint64_t w[4], sh[4];
    memcpy(w, omega_ext, 32);
    if (coset_shift) memcpy(sh, coset_shift, 32);
    if (form == BZH_FORM_CANONICAL) {
        int rc = host_to_montgomery(field, w);
        if (!rc && coset_shift) rc = host_to_montgomery(field, sh);
        if (rc) return rc;
    }
    int rc = ntt_run_padded(ctx, field, d_out, d_in, log_n, log_ext, batch, w, coset_shift ? sh : nullptr);
    if (rc) return rc;
    if (form == BZH_FORM_CANONICAL) {
        rc = field_convert(ctx, field, d_out, out_elems, 0);
        if (rc) return rc;
    }
    if (mem == BZH_MEM_HOST) {
        BZH_HIP_TRY(ctx, hipMemcpyAsync(out, d_out, out_elems * 32, hipMemcpyDeviceToHost, ctx->stream));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return BZH_OK;
}

#define BZH_POLY_PROLOGUE(cond_args)                                                            \
    if (!ctx || !valid_field(field) || !valid_form(form) || !valid_mem(mem) || (cond_args)) return BZH_E_ARG; \
    std::lock_guard<std::mutex> lk(ctx->mu);                                                    \
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));                                                \
    int rc = BZH_OK;                                                                            \
    (void)rc

int bzh_field_convert(bzh_ctx* ctx, int field, uint64_t* data, size_t count, int to_montgomery, int mem) {
    const int form = BZH_FORM_MONTGOMERY;
    BZH_POLY_PROLOGUE(!data && count);
    if (!count) return BZH_OK;
    if (mem == BZH_MEM_DEVICE) return field_convert(ctx, field, (uint32_t*)data, count, to_montgomery ? 1 : 0);
    Stager s{ctx, field, BZH_FORM_MONTGOMERY};
    uint32_t* d;
    if ((rc = s.begin(count * 32))) return rc;
    if ((rc = s.in(data, count, &d))) return rc;
    if ((rc = field_convert(ctx, field, d, count, to_montgomery ? 1 : 0))) return rc;
    return s.out(data, d, count);
}

int bzh_random_field(bzh_ctx* ctx, int field, const uint8_t* rng_bytes, size_t count, int form, int mem, uint64_t* out) {
    BZH_POLY_PROLOGUE((!rng_bytes || !out) && count);
    if (!count) return BZH_OK;
    Stager s{ctx, field, BZH_FORM_MONTGOMERY};
    if ((rc = s.begin(count * 64 + (mem == BZH_MEM_HOST ? count * 32 : 0) + 256))) return rc;
    uint32_t* d_raw = s.carve(count * 64);
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_raw, rng_bytes, count * 64, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* d_out = mem == BZH_MEM_HOST ? s.carve(count * 32) : (uint32_t*)out;
    if ((rc = random_field(ctx, field, d_raw, count, d_out))) return rc;
    if (form == BZH_FORM_CANONICAL && (rc = field_convert(ctx, field, d_out, count, 0))) return rc;
    if (mem == BZH_MEM_HOST) {
        Stager so{ctx, field, BZH_FORM_MONTGOMERY};  // already in the requested form: plain copy out
        return so.out(out, d_out, count);
    }
    return BZH_OK;  // staging reuse by the next call is stream-ordered
}

int bzh_batch_invert(bzh_ctx* ctx, int field, uint64_t* data, size_t count, int form, int mem) {
    BZH_POLY_PROLOGUE(!data && count);
    if (!count) return BZH_OK;
    if (mem == BZH_MEM_DEVICE) {
        uint32_t* d = (uint32_t*)data;
        if (form == BZH_FORM_CANONICAL && (rc = field_convert(ctx, field, d, count, 1))) return rc;
        if ((rc = poly_batch_invert(ctx, field, d, count))) return rc;
        if (form == BZH_FORM_CANONICAL) rc = field_convert(ctx, field, d, count, 0);
        return rc;
    }
    Stager s{ctx, field, form};
    uint32_t* d;
    if ((rc = s.begin(count * 32))) return rc;
    if ((rc = s.in(data, count, &d))) return rc;
    if ((rc = poly_batch_invert(ctx, field, d, count))) return rc;
    return s.out(data, d, count);
}

int bzh_prefix_product(bzh_ctx* ctx, int field, uint64_t* data, size_t n, size_t batch, int form, int mem) {
    BZH_POLY_PROLOGUE(!data && n && batch);
    const size_t count = n * batch;
    if (!count) return BZH_OK;
    if (mem == BZH_MEM_DEVICE) {
        uint32_t* d = (uint32_t*)data;
        if (form == BZH_FORM_CANONICAL && (rc = field_convert(ctx, field, d, count, 1))) return rc;
        if ((rc = poly_prefix_product(ctx, field, d, n, batch))) return rc;
        if (form == BZH_FORM_CANONICAL) rc = field_convert(ctx, field, d, count, 0);
        return rc;
    }
    Stager s{ctx, field, form};
    uint32_t* d;
    if ((rc = s.begin(count * 32))) return rc;
    if ((rc = s.in(data, count, &d))) return rc;
    if ((rc = poly_prefix_product(ctx, field, d, n, batch))) return rc;
    return s.out(data, d, count);
}

int bzh_eval_polynomial(bzh_ctx* ctx, int field, const uint64_t* coeffs, size_t n, size_t batch, const uint64_t* xs, size_t nx,
                        int form, int mem, uint64_t* out) {
    BZH_POLY_PROLOGUE(!coeffs || !xs || !out || !n || (nx != 1 && nx != batch));
    if (!batch) return BZH_OK;
    const size_t stride = nx == 1 ? 0 : 1;
    if (mem == BZH_MEM_DEVICE) {
        if (form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
        return poly_eval(ctx, field, (const uint32_t*)coeffs, n, batch, (const uint32_t*)xs, stride, (uint32_t*)out);
    }
    Stager s{ctx, field, form};
    uint32_t *dc, *dx;
    if ((rc = s.begin((n * batch + nx + batch) * 32 + 256))) return rc;
    if ((rc = s.in(coeffs, n * batch, &dc))) return rc;
    if ((rc = s.in(xs, nx, &dx))) return rc;
    uint32_t* dout = s.carve(batch * 32);
    if ((rc = poly_eval(ctx, field, dc, n, batch, dx, stride, dout))) return rc;
    return s.out(out, dout, batch);
}

int bzh_inner_product(bzh_ctx* ctx, int field, const uint64_t* a, const uint64_t* b, size_t n, size_t batch, int form, int mem,
                      uint64_t* out) {
    BZH_POLY_PROLOGUE(!a || !b || !out || !n);
    if (!batch) return BZH_OK;
    if (mem == BZH_MEM_DEVICE) {
        if (form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
        return poly_inner_product(ctx, field, (const uint32_t*)a, (const uint32_t*)b, n, batch, (uint32_t*)out);
    }
    Stager s{ctx, field, form};
    uint32_t *da, *db;
    if ((rc = s.begin((2 * n * batch + batch) * 32 + 256))) return rc;
    if ((rc = s.in(a, n * batch, &da))) return rc;
    if ((rc = s.in(b, n * batch, &db))) return rc;
    uint32_t* dout = s.carve(batch * 32);
    if ((rc = poly_inner_product(ctx, field, da, db, n, batch, dout))) return rc;
    return s.out(out, dout, batch);
}

int bzh_fold(bzh_ctx* ctx, int field, const uint64_t* in, size_t half, size_t batch, const uint64_t* u, size_t nu, int form,
             int mem, uint64_t* out) {
    BZH_POLY_PROLOGUE(!in || !u || !out || (nu != 1 && nu != batch));
    if (!batch || !half) return BZH_OK;
    const size_t stride = nu == 1 ? 0 : 1;
    if (mem == BZH_MEM_DEVICE) {
        if (form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
        return poly_fold(ctx, field, (const uint32_t*)in, half, batch, (const uint32_t*)u, stride, (uint32_t*)out);
    }
    Stager s{ctx, field, form};
    uint32_t *din, *du;
    if ((rc = s.begin((3 * half * batch + nu) * 32 + 256))) return rc;
    if ((rc = s.in(in, 2 * half * batch, &din))) return rc;
    if ((rc = s.in(u, nu, &du))) return rc;
    uint32_t* dout = s.carve(half * batch * 32);
    if ((rc = poly_fold(ctx, field, din, half, batch, du, stride, dout))) return rc;
    return s.out(out, dout, half * batch);
}

int bzh_vec_mul(bzh_ctx* ctx, int field, uint64_t* a, const uint64_t* b, size_t count, int form, int mem) {
    BZH_POLY_PROLOGUE((!a || !b) && count);
    if (!count) return BZH_OK;
    if (mem == BZH_MEM_DEVICE) {
        if (form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
        return poly_vec_mul(ctx, field, (uint32_t*)a, (const uint32_t*)b, count);
    }
    Stager s{ctx, field, form};
    uint32_t *da, *db;
    if ((rc = s.begin(2 * count * 32 + 256))) return rc;
    if ((rc = s.in(a, count, &da))) return rc;
    if ((rc = s.in(b, count, &db))) return rc;
    if ((rc = poly_vec_mul(ctx, field, da, db, count))) return rc;
    return s.out(a, da, count);
}

int bzh_kate_division_batch(bzh_ctx* ctx, int field, const uint64_t* coeffs, size_t n, size_t batch, const uint64_t* xs, int form,
                            int mem, uint64_t* out) {
    BZH_POLY_PROLOGUE(!coeffs || !xs || !out || n < 1 || !batch || batch > 65535);
    if (n == 1) return BZH_OK;
    if (mem == BZH_MEM_DEVICE && form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
    // x_v per vector, uploaded in Montgomery form
    std::vector<uint64_t> xh(batch * 4);
    auto go = [&](auto tag) {
        using PP = decltype(tag);
        for (size_t v = 0; v < batch; v++) store_host<PP>(xh.data() + v * 4, load_host<PP>(xs + 4 * v, form), BZH_FORM_MONTGOMERY);
    };
    switch (field) {
        case BZH_FIELD_FP: go(FpParams{}); break;
        case BZH_FIELD_FQ: go(FqParams{}); break;
        case BZH_FIELD_BN254_FR: go(BnFrParams{}); break;
        default: go(BnFqParams{}); break;
    }
    Stager s{ctx, field, BZH_FORM_MONTGOMERY};
    if ((rc = s.begin(((mem == BZH_MEM_HOST ? 2 * n : 0) + 1) * batch * 32 + 512))) return rc;
    uint32_t* d_x;
    if ((rc = s.in(xh.data(), batch, &d_x))) return rc;
    if (mem == BZH_MEM_HOST) {
        Stager sc{ctx, field, form};
        sc.cur = s.cur;
        uint32_t* d_c;
        if ((rc = sc.in(coeffs, n * batch, &d_c))) return rc;
        uint32_t* d_q = sc.carve((n - 1) * batch * 32);
        if ((rc = poly_kate_division(ctx, field, d_c, n, batch, d_x, d_q))) return rc;
        return sc.out(out, d_q, (n - 1) * batch);
    }
    rc = poly_kate_division(ctx, field, (const uint32_t*)coeffs, n, batch, d_x, (uint32_t*)out);
    return rc;  // staging reuse by the next call is stream-ordered
}

int bzh_kate_division(bzh_ctx* ctx, int field, const uint64_t* coeffs, size_t n, const uint64_t* x, int form, int mem,
                      uint64_t* out) {
    return bzh_kate_division_batch(ctx, field, coeffs, n, 1, x, form, mem, out);
}

int bzh_permute_expression_pair(int field, const uint64_t* input, const uint64_t* table, size_t usable_rows, int form,
                                uint64_t* out_input, uint64_t* out_table) {
    if (!valid_field(field) || !valid_form(form) || ((!input || !table || !out_input || !out_table) && usable_rows)) return BZH_E_ARG;
    switch (field) {
        case BZH_FIELD_FP: return permute_pair_host<FpParams>(input, table, usable_rows, form, out_input, out_table);
        case BZH_FIELD_FQ: return permute_pair_host<FqParams>(input, table, usable_rows, form, out_input, out_table);
        case BZH_FIELD_BN254_FR: return permute_pair_host<BnFrParams>(input, table, usable_rows, form, out_input, out_table);
        default: return permute_pair_host<BnFqParams>(input, table, usable_rows, form, out_input, out_table);
    }
}

int bzh_expr_eval_batch(bzh_ctx* ctx, int field, const bzh_expr_op* prog, size_t nops, const uint64_t* const* columns,
                        const size_t* column_strides, size_t ncols, const uint64_t* consts, size_t nconsts, size_t const_stride,
                        unsigned log_size, int result_slot, size_t batch, int form, int mem, uint64_t* out) {
    BZH_POLY_PROLOGUE(!prog || !nops || (!columns && ncols) || (!consts && nconsts) || !out || log_size > 30 ||
                      result_slot < 0 || result_slot >= BZH_EXPR_MAX_SLOTS || !batch || batch > 65535 ||
                      (const_stride && const_stride < nconsts));
    const size_t size = (size_t)1 << log_size;
    int nslots = result_slot + 1;
    for (size_t i = 0; i < nops; i++) {  // validate the program: it indexes device memory
        const bzh_expr_op& o = prog[i];
        if (o.op > BZH_EXPR_COPY || o.dst >= BZH_EXPR_MAX_SLOTS) return BZH_E_ARG;
        nslots = std::max(nslots, (int)o.dst + 1);
        const int kinds[2] = {o.a_kind, (o.op == BZH_EXPR_NEG || o.op == BZH_EXPR_COPY) ? BZH_EXPR_SLOT : o.b_kind};
        const int idxs[2] = {o.a_idx, (o.op == BZH_EXPR_NEG || o.op == BZH_EXPR_COPY) ? 0 : o.b_idx};
        for (int q = 0; q < 2; q++) {
            if (kinds[q] == BZH_EXPR_SLOT && (idxs[q] < 0 || idxs[q] >= BZH_EXPR_MAX_SLOTS)) return BZH_E_ARG;
            if (kinds[q] == BZH_EXPR_SLOT) nslots = std::max(nslots, idxs[q] + 1);
            if (kinds[q] == BZH_EXPR_COLUMN && (idxs[q] < 0 || (size_t)idxs[q] >= ncols)) return BZH_E_ARG;
            if (kinds[q] == BZH_EXPR_CONST && (idxs[q] < 0 || (size_t)idxs[q] >= nconsts)) return BZH_E_ARG;
            if (kinds[q] > BZH_EXPR_CONST || kinds[q] < 0) return BZH_E_ARG;
        }
    }
    if (mem == BZH_MEM_DEVICE && form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
    // host columns: vector v of column c is columns[c] + v * stride * 4 limbs; they are staged as `reps` consecutive copies
    std::vector<size_t> strides(ncols, 0), reps(ncols, 1);
    size_t host_elems = 0;
    for (size_t c = 0; c < ncols; c++) {
        strides[c] = (column_strides && batch > 1) ? column_strides[c] : 0;
        if (strides[c] && strides[c] < size) return BZH_E_ARG;
        reps[c] = strides[c] ? batch : 1;
        host_elems += reps[c] * size;
    }
    const size_t nconst_total = const_stride ? const_stride * (batch - 1) + nconsts : nconsts;
    Stager s{ctx, field, form};
    if ((rc = s.begin(((mem == BZH_MEM_HOST ? host_elems + batch * size : 0) + nconst_total) * 32 + nops * sizeof(bzh_expr_op) +
                      ncols * (sizeof(void*) + sizeof(size_t) + 128) + 1024)))
        return rc;
    std::vector<const uint32_t*> ptrs(ncols);
    for (size_t c = 0; c < ncols; c++) {
        if (mem == BZH_MEM_HOST) {
            uint32_t* d = s.carve(reps[c] * size * 32);
            for (size_t v = 0; v < reps[c]; v++)
                BZH_HIP_TRY(ctx, hipMemcpyAsync(d + v * size * 8, columns[c] + v * strides[c] * 4, size * 32, hipMemcpyHostToDevice,
                                                ctx->stream));
            if (form == BZH_FORM_CANONICAL && (rc = field_convert(ctx, field, d, reps[c] * size, 1))) return rc;
            ptrs[c] = d;
            if (strides[c]) strides[c] = size;
        } else {
            ptrs[c] = (const uint32_t*)columns[c];
        }
    }
    uint32_t* d_consts;
    if ((rc = s.in(consts, nconst_total, &d_consts))) return rc;
    uint32_t* d_prog = s.carve(nops * sizeof(bzh_expr_op));
    uint32_t* d_ptrs = s.carve(ncols * sizeof(void*) + 8);
    uint32_t* d_strides = s.carve(ncols * sizeof(size_t) + 8);
    if ((rc = h2d_small(ctx, d_prog, prog, nops * sizeof(bzh_expr_op)))) return rc;
    if (ncols && (rc = h2d_small(ctx, d_ptrs, ptrs.data(), ncols * sizeof(void*)))) return rc;
    if (ncols && (rc = h2d_small(ctx, d_strides, strides.data(), ncols * sizeof(size_t)))) return rc;
    uint32_t* d_out = mem == BZH_MEM_HOST ? s.carve(batch * size * 32) : (uint32_t*)out;
    rc = expr_eval(ctx, field, d_prog, (int)nops, (const uint32_t* const*)d_ptrs, (const size_t*)d_strides, d_consts, const_stride, size,
                   result_slot, batch, nslots, d_out);
    if (rc) return rc;
    if (mem == BZH_MEM_HOST) return s.out(out, d_out, batch * size);
    return BZH_OK;  // staging reuse by the next call is stream-ordered
}

int bzh_expr_eval(bzh_ctx* ctx, int field, const bzh_expr_op* prog, size_t nops, const uint64_t* const* columns, size_t ncols,
                  const uint64_t* consts, size_t nconsts, unsigned log_size, int result_slot, int form, int mem, uint64_t* out) {
    return bzh_expr_eval_batch(ctx, field, prog, nops, columns, nullptr, ncols, consts, nconsts, 0, log_size, result_slot, 1, form, mem,
                               out);
}

int bzh_ipa_open_batch(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* polys, int form, int mem, size_t batch,
                       const uint64_t* blinds, const uint64_t* x3s, const uint8_t* rng, size_t rng_stride,
                       bzh_transcript* const* transcripts, uint64_t* out_v) {
    if (!ctx || !bases || !polys || !blinds || !x3s || !rng || !transcripts || !out_v || !valid_form(form) || !valid_mem(mem))
        return BZH_E_ARG;
    if (bases->n < 3 || bases->device != ctx->device || !batch || batch > 32767) return BZH_E_ARG;
    const size_t n = bases->n - 2;
    if (n & (n - 1)) return BZH_E_ARG;
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    if (rng_stride < 64 * (n + 1 + 2 * (size_t)k)) return BZH_E_ARG;
    for (size_t b = 0; b < batch; b++)
        if (!transcripts[b]) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int field = bases->curve == BZH_CURVE_VESTA ? BZH_FIELD_FP : (bases->curve == BZH_CURVE_PALLAS ? BZH_FIELD_FQ : BZH_FIELD_BN254_FR);
    if (mem == BZH_MEM_DEVICE) {
        if (form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
        return ipa_open(ctx, bases, (const uint32_t*)polys, batch, blinds, x3s, rng, rng_stride, transcripts, out_v);
    }
    int rc;
    Stager s{ctx, field, form};  // workspace slot 3; the opening itself works in slots 0-2, 4 and 5
    if ((rc = s.begin(batch * n * 32))) return rc;
    uint32_t* d_polys;
    if ((rc = s.in(polys, batch * n, &d_polys))) return rc;
    return ipa_open(ctx, bases, d_polys, batch, blinds, x3s, rng, rng_stride, transcripts, out_v);
}

int bzh_ipa_open(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* poly, int form, int mem, const uint64_t* blind,
                 const uint64_t* x3, const uint8_t* rng, size_t rng_len, bzh_transcript* transcript, uint64_t* out_v) {
    return bzh_ipa_open_batch(ctx, bases, poly, form, mem, 1, blind, x3, rng, rng_len, &transcript, out_v);
}

int bzh_ipa_verify(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* commitment_xy, const uint64_t* x3, const uint64_t* v,
                   const uint8_t* proof, size_t proof_len, bzh_transcript* transcript, const uint64_t* g0_u_w) {
    if (!ctx || !bases || !commitment_xy || !x3 || !v || !proof || !transcript || !g0_u_w) return BZH_E_ARG;
    if (bases->n < 3 || bases->device != ctx->device) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ipa_verify(ctx, bases, commitment_xy, x3, v, proof, proof_len, transcript, g0_u_w);
}

int bzh_jacobian_to_affine(int curve, const uint64_t* xyz, size_t n, int form, uint64_t* out_xy) {
    if ((!xyz || !out_xy) && n) return BZH_E_ARG;
    if (!valid_curve(curve) || !valid_form(form)) return BZH_E_ARG;
    switch (curve) {
        case BZH_CURVE_VESTA: jac_to_aff_host<FqParams>(xyz, n, form, out_xy); break;
        case BZH_CURVE_PALLAS: jac_to_aff_host<FpParams>(xyz, n, form, out_xy); break;
        case BZH_CURVE_BN254: jac_to_aff_host<BnFqParams>(xyz, n, form, out_xy); break;
    }
    return BZH_OK;
}

int bzh_jacobian_sum(int curve, const uint64_t* xyz, size_t n, int form, uint64_t* out_xyz) {
    if (!out_xyz || (!xyz && n)) return BZH_E_ARG;
    if (!valid_curve(curve) || !valid_form(form)) return BZH_E_ARG;
    switch (curve) {
        case BZH_CURVE_VESTA: jac_sum_host<FqParams>(xyz, n, form, out_xyz); break;
        case BZH_CURVE_PALLAS: jac_sum_host<FpParams>(xyz, n, form, out_xyz); break;
        case BZH_CURVE_BN254: jac_sum_host<BnFqParams>(xyz, n, form, out_xyz); break;
    }
    return BZH_OK;
}

int bzh_affine_compress(int curve, const uint64_t* xy, size_t n, int form, uint8_t* out32) {
    if ((!xy || !out32) && n) return BZH_E_ARG;
    if (!valid_curve(curve) || !valid_form(form)) return BZH_E_ARG;
    switch (curve) {
        case BZH_CURVE_VESTA: compress_host<FqParams>(xy, n, form, out32); break;
        case BZH_CURVE_PALLAS: compress_host<FpParams>(xy, n, form, out32); break;
        case BZH_CURVE_BN254: compress_host<BnFqParams>(xy, n, form, out32); break;
    }
    return BZH_OK;
}

int bzh_field_omega(int field, unsigned log_n, int form, uint64_t* out) {
    if (!out || !valid_field(field) || !valid_form(form)) return BZH_E_ARG;
    if (log_n > field_two_adicity(field)) return BZH_E_RANGE;
    switch (field) {
        case BZH_FIELD_FP: omega_host<FpParams>(32, 5, log_n, form, out); break;
        case BZH_FIELD_FQ: omega_host<FqParams>(32, 5, log_n, form, out); break;
        case BZH_FIELD_BN254_FR: omega_host<BnFrParams>(28, 7, log_n, form, out); break;
        case BZH_FIELD_BN254_FQ: omega_host<BnFqParams>(1, 3, log_n, form, out); break;
    }
    return BZH_OK;
}


}  // extern "C"
