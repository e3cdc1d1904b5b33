// rng.hpp — randomness of the CPU client (key generation, encryption masks and noise).
//
// Two modes, chosen by the `seed` argument of helm_client_keygen / helm_si_client_keygen:
//
//   seed == HELM_SEED_OS_ENTROPY (0)   the default of every public entry point.  Every stream is a
//       ChaCha20 keystream (RFC 8439 block function, 20 rounds) under a 256-bit key drawn from the
//       operating system (getrandom(2)); the stream number goes into the nonce.  Encryption
//       randomness gets its OWN key from the OS, independent of the one that produced the secret
//       key.  This is what tfhe::boolean::gen_keys() / shortint::gen_keys() do in the reference
//       (OS-seeded CSPRNG, reference src/bin/helm.rs:241,301).
//
//   seed != 0   DETERMINISTIC, INSECURE: xoshiro256** streams derived from the 64-bit seed, for tests,
//       golden vectors and benchmarks that must reproduce (the reference's tests do the same with a
//       fixed seed, tests/circuit_test.rs:119).  Ciphertext masks are raw generator outputs and a
//       64-bit seed can be searched: never use a fixed seed for data that matters.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>

#include <sys/random.h>

namespace helm_rng {

constexpr uint64_t SEED_OS_ENTROPY = 0;

struct OsKey {
    uint32_t w[8];
    static OsKey draw()
    {
        OsKey k;
        unsigned char *p = reinterpret_cast<unsigned char *>(k.w);
        size_t got = 0;
        while (got < sizeof(k.w)) {
            const ssize_t r = getrandom(p + got, sizeof(k.w) - got, 0);
            if (r <= 0) throw std::runtime_error("getrandom() failed: no OS entropy for key generation");
            got += (size_t)r;
        }
        return k;
    }
};

class Rng {
    // ---- deterministic mode: xoshiro256** ------------------------------------------------
    uint64_t s[4] = {0, 0, 0, 0};
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    // ---- secure mode: ChaCha20 keystream -------------------------------------------------
    bool secure = false;
    uint32_t key[8] = {0}, nonce[2] = {0, 0};
    uint64_t counter = 0;
    uint64_t buf[8];
    int buf_pos = 8;
    static uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    void refill()
    {
        uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                          key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), nonce[0], nonce[1]};
        uint32_t in[16];
        std::memcpy(in, x, sizeof(x));
#define HELM_QR(a, b, c, d)                                                                                    \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);                \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
        for (int r = 0; r < 10; r++) {
            HELM_QR(0, 4, 8, 12) HELM_QR(1, 5, 9, 13) HELM_QR(2, 6, 10, 14) HELM_QR(3, 7, 11, 15)
            HELM_QR(0, 5, 10, 15) HELM_QR(1, 6, 11, 12) HELM_QR(2, 7, 8, 13) HELM_QR(3, 4, 9, 14)
        }
#undef HELM_QR
        for (int i = 0; i < 16; i++) x[i] += in[i];
        std::memcpy(buf, x, sizeof(buf));
        counter++;
        buf_pos = 0;
    }

    bool have_spare = false;
    double spare = 0;

public:
    Rng() = default;
    // deterministic (insecure) stream `stream` of `seed`
    Rng(uint64_t seed, uint64_t stream)
    {
        uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
        for (auto &v : s) v = splitmix(x);
    }
    // ChaCha20 stream `stream` under an OS-drawn key
    Rng(const OsKey &k, uint64_t stream) : secure(true)
    {
        std::memcpy(key, k.w, sizeof(key));
        nonce[0] = (uint32_t)stream;
        nonce[1] = (uint32_t)(stream >> 32);
    }
    ~Rng()
    {
        volatile uint32_t *p = key;
        for (int i = 0; i < 8; i++) p[i] = 0;
    }
    bool is_secure() const { return secure; }
    // RFC 8439 section 2.3.2 block-function test vector (key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a
    // 00:00:00:00 - its 32-bit counter and first nonce word are this layout's 64-bit counter)
    static bool selftest()
    {
        OsKey k;
        for (int i = 0; i < 8; i++) k.w[i] = (uint32_t)(4 * i) | (uint32_t)(4 * i + 1) << 8 | (uint32_t)(4 * i + 2) << 16 | (uint32_t)(4 * i + 3) << 24;
        Rng r(k, 0x4a000000ull);
        r.counter = 1ull | (0x09000000ull << 32);
        static const uint32_t want[16] = {0xe4e7f110u, 0x15593bd1u, 0x1fdd0f50u, 0xc47120a3u, 0xc7f4d1c7u, 0x0368c033u,
                                          0x9aaa2204u, 0x4e6cd4c3u, 0x466482d2u, 0x09aa9f07u, 0x05d7c214u, 0xa2028bd9u,
                                          0xd19c12b5u, 0xb94e16deu, 0xe883d0cbu, 0x4e3c50a2u};
        for (int i = 0; i < 8; i++) {
            const uint64_t v = r.next();
            if ((uint32_t)v != want[2 * i] || (uint32_t)(v >> 32) != want[2 * i + 1]) return false;
        }
        return true;
    }

    uint64_t next()
    {
        if (secure) {
            if (buf_pos == 8) refill();
            return buf[buf_pos++];
        }
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    uint32_t u32() { return (uint32_t)(next() >> 32); }
    double unit() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); } // (0,1)
    double gauss()
    {
        if (have_spare) {
            have_spare = false;
            return spare;
        }
        const double u = unit(), v = unit();
        const double r = std::sqrt(-2.0 * std::log(u)), a = 6.283185307179586476925 * v;
        spare = r * std::sin(a);
        have_spare = true;
        return r * std::cos(a);
    }
    // torus noise: round(gauss * std * 2^w) mod 2^w
    uint32_t noise32(double std_dev) { return (uint32_t)(int64_t)std::llround(gauss() * std_dev * 4294967296.0); }
    uint64_t noise64(double std_dev) { return (uint64_t)(int64_t)std::llround(gauss() * std_dev * 18446744073709551616.0); }
};

// The streams of one key generation: deterministic from `seed`, or ChaCha20 under one OS-drawn key;
// `encryption()` is the generator the client key keeps for its ciphertexts (its own OS key in secure mode).
struct Source {
    uint64_t seed;
    OsKey os{};
    explicit Source(uint64_t seed_) : seed(seed_)
    {
        if (seed == SEED_OS_ENTROPY) os = OsKey::draw();
    }
    ~Source()
    {
        volatile uint32_t *p = os.w;
        for (int i = 0; i < 8; i++) p[i] = 0;
    }
    Rng stream(uint64_t id) const { return seed == SEED_OS_ENTROPY ? Rng(os, id) : Rng(seed, id); }
    Rng encryption(uint64_t id) const { return seed == SEED_OS_ENTROPY ? Rng(OsKey::draw(), 0) : Rng(seed, id); }
};

} // namespace helm_rng
