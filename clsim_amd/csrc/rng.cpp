// Host set-up of the per-step multiply-with-carry streams.
//
// The reference reads its multipliers from a "safeprimes" file that is not part
// of the repository (downloaded at build time, CMakeLists.txt:262-271) and was
// produced by private/make_safeprimes/main.cxx:31-104: descending from
// a = 4294967118, keep a when n2 = a*2^32 - 1 and n1 = (n2 - 1)/2 are both
// prime.  That search is redone here: a sieve over small primes removes most
// candidates (both n2 and n1 are linear in a), a deterministic Miller-Rabin
// test on 64-bit integers decides the rest.  State words follow the validity
// loop of private/opencl/mwcrng_init.h:105-113.
#include <algorithm>
#include <atomic>
#include <thread>

#include "host_model.h"

namespace clsimhip {
namespace {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return static_cast<uint64_t>((static_cast<u128>(a) * b) % m); }

uint64_t powmod(uint64_t b, uint64_t e, uint64_t m)
{
    uint64_t r = 1;
    b %= m;
    while (e) {
        if (e & 1) r = mulmod(r, b, m);
        b = mulmod(b, b, m);
        e >>= 1;
    }
    return r;
}

// deterministic for every 64-bit n with these 12 bases
bool is_prime_u64(uint64_t n)
{
    static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (uint64_t p : bases) {
        if (n % p == 0) return n == p;
    }
    uint64_t d = n - 1;
    int r = 0;
    while ((d & 1) == 0) { d >>= 1; ++r; }
    for (uint64_t a : bases) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int i = 1; i < r; ++i) {
            x = mulmod(x, x, n);
            if (x == n - 1) { composite = false; break; }
        }
        if (composite) return false;
    }
    return true;
}

struct SievePrime { uint32_t p, r2, r1; };  // a == r2 (mod p) kills n2, a == r1 kills n1

std::vector<SievePrime> sieve_primes(uint32_t limit)
{
    std::vector<bool> composite(limit + 1, false);
    std::vector<SievePrime> out;
    for (uint32_t p = 2; p <= limit; ++p) {
        if (composite[p]) continue;
        for (uint64_t q = static_cast<uint64_t>(p) * p; q <= limit; q += p) composite[q] = true;
        if (p == 2) continue;   // n2 and n1 are odd for every a
        // a * 2^32 == 1 (mod p)  <=>  a == (2^32)^-1 ; inverse by Fermat
        const uint64_t inv32 = powmod(powmod(2, 32, p), p - 2, p);
        const uint64_t inv31 = powmod(powmod(2, 31, p), p - 2, p);
        out.push_back({p, static_cast<uint32_t>(inv32), static_cast<uint32_t>(inv31)});
    }
    return out;
}

// multipliers in (hi - len, hi], descending
void scan_segment(const std::vector<SievePrime> &primes, uint32_t hi, uint32_t len, std::vector<uint32_t> &out)
{
    std::vector<uint8_t> dead(len, 0);   // index k <-> a = hi - k
    for (const SievePrime &sp : primes) {
        for (uint32_t r : {sp.r2, sp.r1}) {
            // smallest k >= 0 with (hi - k) % p == r
            const uint32_t k0 = static_cast<uint32_t>((static_cast<uint64_t>(hi % sp.p) + sp.p - r) % sp.p);
            for (uint64_t k = k0; k < len; k += sp.p) dead[k] = 1;
        }
    }
    for (uint32_t k = 0; k < len; ++k) {
        if (dead[k]) continue;
        const uint64_t a = static_cast<uint64_t>(hi) - k;
        const uint64_t n2 = (a << 32) - 1;
        if (!is_prime_u64(n2)) continue;
        if (!is_prime_u64((n2 - 1) >> 1)) continue;
        out.push_back(static_cast<uint32_t>(a));
    }
}

inline uint64_t splitmix64(uint64_t &state)
{
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

} // namespace

void mwc_multipliers(uint32_t *out, size_t count)
{
    if (count == 0) return;
    static const std::vector<SievePrime> primes = sieve_primes(30000);
    const uint32_t seg = 1u << 20;
    const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    uint32_t hi = 4294967118u;
    size_t have = 0;
    while (have < count) {
        // one wave of segments, scanned in parallel, consumed in descending order
        std::vector<std::vector<uint32_t>> found(nthreads);
        std::vector<std::thread> pool;
        std::vector<uint32_t> seg_hi(nthreads), seg_len(nthreads);
        unsigned used = 0;
        uint32_t cursor = hi;
        for (unsigned t = 0; t < nthreads && cursor > 2; ++t) {
            const uint32_t len = std::min(seg, cursor - 2);
            seg_hi[t] = cursor; seg_len[t] = len;
            cursor -= len;
            ++used;
        }
        if (used == 0) throw Error(CLSIMHIP_ERR_CONFIG, "ran out of 32-bit MWC multipliers");
        for (unsigned t = 0; t < used; ++t)
            pool.emplace_back([&, t] { scan_segment(primes, seg_hi[t], seg_len[t], found[t]); });
        for (auto &th : pool) th.join();
        for (unsigned t = 0; t < used && have < count; ++t)
            for (uint32_t a : found[t]) {
                if (have >= count) break;
                out[have++] = a;
            }
        hi = cursor;
    }
}

// mwcrng_init.h:105-113; I3RandomService::Integer(0xffffffff) is supplied by
// splitmix64(seed): Integer() := (next() >> 32) % 0xffffffff  (the random
// service belongs to phys-services, outside clsim; callers with their own
// service use clsimhip_initialize_with_streams).
void seed_streams(const uint32_t *a, size_t count, uint64_t seed, uint64_t *x)
{
    uint64_t state = seed;
    for (size_t i = 0; i < count; ++i) {
        uint64_t xi = 0;
        while ((xi == 0) | ((static_cast<uint32_t>(xi >> 32)) >= (a[i] - 1)) | ((static_cast<uint32_t>(xi)) >= 0xfffffffful)) {
            xi = static_cast<uint32_t>((splitmix64(state) >> 32) % 0xffffffffull);
            xi = xi << 32;
            xi += static_cast<uint32_t>((splitmix64(state) >> 32) % 0xffffffffull);
        }
        x[i] = xi;
    }
}

} // namespace clsimhip
