// rrng.hpp -- R's default random number stream, needed because the projector must be
// bit-faithful to the reference's set.seed()/sample() calls (R/ranM.R:20-30,
// R/SHARP.R:493-499).  R-internal algorithms (not vendored in the reference):
// Mersenne-Twister with set.seed()'s LCG scrambling, unif_rand() fix-up, and the
// R >= 3.6 "Rejection" integer sampler.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace sharp {

class RRng {
public:
    explicit RRng(uint32_t seed) { set_seed(seed); }

    void set_seed(uint32_t seed) {
        // Randomize(kind): 50 warm-up steps of x <- 69069 x + 1, then one step per state
        // word; word 0 is the position counter, which FixupSeeds forces to 624.
        for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;
        seed = 69069u * seed + 1u;  // would be the position counter
        for (int j = 0; j < kN; ++j) {
            seed = 69069u * seed + 1u;
            state_[j] = seed;
        }
        pos_ = kN;
    }

    // unif_rand(): tempered 32-bit output scaled into (0,1)
    double unif() {
        if (pos_ >= kN) refill();
        uint32_t y = state_[pos_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        const double v = static_cast<double>(y) * 2.3283064365386963e-10;
        constexpr double kHalfUlp = 0.5 * 2.328306437080797e-10;
        if (v <= 0.0) return kHalfUlp;
        if (1.0 - v <= 0.0) return 1.0 - kHalfUlp;
        return v;
    }

    // R_unif_index(dn): rejection sampling from the next power of two
    double unif_index(double dn) {
        if (dn <= 0) return 0.0;
        const int bits = static_cast<int>(std::ceil(std::log2(dn)));
        double dv;
        do {
            int64_t v = 0;
            for (int n = 0; n <= bits; n += 16) v = 65536 * v + static_cast<int>(std::floor(unif() * 65536));
            if (bits < 64) v &= ((int64_t{1} << bits) - 1);
            dv = static_cast<double>(v);
        } while (dn <= dv);
        return dv;
    }

    // sample(n): 1-based permutation
    std::vector<int> permutation(int n) {
        std::vector<int> pool(n), out(n);
        for (int i = 0; i < n; ++i) pool[i] = i;
        int left = n;
        for (int i = 0; i < n; ++i) {
            const int j = static_cast<int>(unif_index(left));
            out[i] = pool[j] + 1;
            pool[j] = pool[--left];
        }
        return out;
    }

private:
    static constexpr int kN = 624, kM = 397;
    uint32_t state_[kN];
    int pos_ = kN;

    static uint32_t twist(uint32_t hi, uint32_t lo) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    void refill() {
        for (int k = 0; k < kN - kM; ++k) state_[k] = state_[k + kM] ^ twist(state_[k], state_[k + 1]);
        for (int k = kN - kM; k < kN - 1; ++k) state_[k] = state_[k + kM - kN] ^ twist(state_[k], state_[k + 1]);
        state_[kN - 1] = state_[kM - 1] ^ twist(state_[kN - 1], state_[0]);
        pos_ = 0;
    }
};

}  // namespace sharp
