// Round 4, VERDICT r3 next #1(c): ONE PERMUTATION-LANE of the device transcript before any integration.
//
// ligero_amd/csrc/sponge_kernels.h runs the prover's Fiat-Shamir sponge (Poseidon width 3, alpha 17, 8 + 31 rounds: 275 field
// products per permutation, 326 permutations per Poseidon-shape proof) with one lane per proof.  This program
//   1. checks the kernel against a host restatement of the same sponge (absorb(&Vec<u8>) of a digest, squeeze_bytes(32),
//      absorb(&Vec<F>) with and without trailing zeros, two more squeezes; the distinct-index kernel against a host loop),
//   2. times a chain of permutations: microseconds per permutation for 1 ... many waves, alone and beside a VALU-bound
//      kernel on another stream (what the commit's transforms are),
//   3. prints the projection the kill criterion asks for: proofs/s with <= 1024 proofs in flight.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ligero_amd/csrc tools/microbench9.hip -o tools/microbench9
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#include "host_fr.h"
#include "sponge_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace lg;
using lg_host::Fr;

// ---- host restatement (4 x u64 Montgomery, R = 2^256), the shape of ligero_amd/host/transcript.hpp
static Fr h_add(const Fr& a, const Fr& b) {
    Fr r; unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || lg_host::geq(r, lg_host::kP)) r = lg_host::sub_raw(r, lg_host::kP);
    return r;
}
struct HostSponge {
    std::vector<std::array<Fr, 3>> ark;
    uint32_t full = 8, partial = 31;
    Fr s[3] = {{{0, 0, 0, 0}}, {{0, 0, 0, 0}}, {{0, 0, 0, 0}}};
    bool squeezing = false; size_t idx = 0;
    static Fr sbox(const Fr& x) { Fr y = lg_host::mul(x, x); y = lg_host::mul(y, y); y = lg_host::mul(y, y); y = lg_host::mul(y, y); return lg_host::mul(y, x); }
    void permute() {
        for (uint32_t r = 0; r < full + partial; r++) {
            for (int j = 0; j < 3; j++) s[j] = h_add(s[j], ark[r][j]);
            const bool f = r < full / 2 || r >= full / 2 + partial;
            s[0] = sbox(s[0]);
            if (f) { s[1] = sbox(s[1]); s[2] = sbox(s[2]); }
            const Fr n0 = h_add(s[0], s[2]), n1 = h_add(s[0], s[1]), n2 = h_add(s[1], s[2]);
            s[0] = n0; s[1] = n1; s[2] = n2;
        }
    }
    void absorb(const std::vector<Fr>& e) {
        if (e.empty()) return;
        size_t start;
        if (squeezing) { permute(); start = 0; } else { start = idx; if (start == 2) { permute(); start = 0; } }
        size_t pos = 0;
        for (;;) {
            const size_t left = e.size() - pos;
            if (start + left <= 2) { for (size_t i = 0; i < left; i++) s[1 + start + i] = h_add(s[1 + start + i], e[pos + i]); squeezing = false; idx = start + left; return; }
            const size_t take = 2 - start;
            for (size_t i = 0; i < take; i++) s[1 + start + i] = h_add(s[1 + start + i], e[pos + i]);
            permute(); pos += take; start = 0;
        }
    }
    void absorb_bytes(const uint8_t* d, size_t len) {
        std::vector<uint8_t> by(8 + len);
        for (int i = 0; i < 8; i++) by[i] = (uint8_t)((uint64_t)len >> (8 * i));
        memcpy(by.data() + 8, d, len);
        std::vector<Fr> el;
        for (size_t off = 0; off < by.size(); off += 31) {
            const size_t take = std::min<size_t>(31, by.size() - off);
            Fr c = {{0, 0, 0, 0}};
            for (size_t i = 0; i < take; i++) c.l[i / 8] |= (uint64_t)by[off + i] << (8 * (i % 8));
            el.push_back(lg_host::to_mont(c));
        }
        absorb(el);
    }
    void squeeze_seed(uint8_t out[32]) {
        size_t start;
        if (!squeezing) { permute(); start = 0; } else { start = idx; if (start == 2) { permute(); start = 0; } }
        Fr e[2];
        if (start == 0) { e[0] = s[1]; e[1] = s[2]; idx = 2; } else { e[0] = s[2]; e[1] = s[1]; idx = 1; }
        squeezing = true;
        std::vector<uint8_t> by;
        for (int k = 0; k < 2; k++) { const Fr c = lg_host::from_mont(e[k]); for (int i = 0; i < 31; i++) by.push_back((uint8_t)(c.l[i / 8] >> (8 * (i % 8)))); }
        memcpy(out, by.data(), 32);
    }
};
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t next64() { uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static Fr rand_fr() { Fr c = {{next64(), next64(), next64(), next64() >> 3}}; return lg_host::to_mont(c); }   // < 2^253 < p
static void to29_261(const Fr& a_mont, uint32_t out[9]) {   // limbs of x * 2^261 mod p
    static const Fr m32 = lg_host::to_mont(Fr{{32, 0, 0, 0}});
    const Fr t = lg_host::mul(a_mont, m32);
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t x = t.l[w] >> sh;
        if (sh > 35 && w < 3) x |= t.l[w + 1] << (64 - sh);
        out[i] = (i < 8) ? (uint32_t)(x & 0x1fffffffu) : (uint32_t)x;
    }
}
static void host_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = s[15] = 0;
    uint32_t x[16]; memcpy(x, s, sizeof(x));
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    auto qr = [&](int a, int b, int c, int d) { x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12); x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7); };
    for (int r = 0; r < 10; r++) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15); qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
static std::vector<uint32_t> host_indices(uint64_t n, uint64_t t, const uint32_t key[8]) {
    std::set<uint64_t> sel;
    const uint64_t to_select = std::min(t, n - t), zone = (n << __builtin_clzll(n)) - 1;
    uint64_t counter = 0; uint32_t buf[16]; int at = 16;
    while (sel.size() < to_select) {
        if (at >= 16) { host_block(key, counter++, buf); at = 0; }
        const uint64_t v = (uint64_t)buf[at] | ((uint64_t)buf[at + 1] << 32); at += 2;
        const unsigned __int128 m = (unsigned __int128)v * n;
        if ((uint64_t)m <= zone) sel.insert((uint64_t)(m >> 64));
    }
    std::vector<uint32_t> out;
    if (to_select == t) for (uint64_t v : sel) out.push_back((uint32_t)v);
    else for (uint64_t i = 0; i < n; i++) if (!sel.count(i)) out.push_back((uint32_t)i);
    return out;
}

// a VALU-bound companion (mul29 chains on every SIMD), to see what a sponge wave gets beside the commit's transforms
__global__ void __launch_bounds__(256) valu_load_kernel(uint32_t* sink, uint32_t iters) {
    f29 x, y;
    for (int i = 0; i < 9; i++) { x.v[i] = (threadIdx.x * 7 + i) & kM29; y.v[i] = (blockIdx.x * 13 + i * 5 + 1) & kM29; }
    x.v[8] &= 0xfffff; y.v[8] &= 0xfffff;
    for (uint32_t i = 0; i < iters; i++) { mul29(x, x, y); mul29(y, y, x); }
    if (x.v[0] == 0xdeadbeefu) sink[0] = y.v[1];
}

// MB9_QUAD=1: the four-lanes-per-proof kernel of round 5 (sponge_quad_kernel: 16 proofs per wave) in the correctness and timing legs
static bool g_quad = false;
#define LAUNCH_SPONGE(B_, stream_, args_) do { \
        if (g_quad) hipLaunchKernelGGL(sponge_quad_kernel, dim3(((B_) + 15) / 16), dim3(64), 0, stream_, args_); \
        else hipLaunchKernelGGL(sponge_kernel<true>, dim3(((B_) + 63) / 64), dim3(64), 0, stream_, args_); } while (0)

int main(int argc, char** argv) {
    g_quad = getenv("MB9_QUAD") && atoi(getenv("MB9_QUAD")) == 1;
    printf("kernel: %s\n", g_quad ? "sponge_quad_kernel (four lanes per proof)" : "sponge_kernel (one lane per proof)");
    const uint32_t FULL = 8, PART = 31, ROUNDS = FULL + PART;
    HostSponge proto;
    proto.ark.resize(ROUNDS);
    std::vector<uint32_t> ark29(27 * ROUNDS);
    for (uint32_t r = 0; r < ROUNDS; r++)
        for (int j = 0; j < 3; j++) { proto.ark[r][j] = rand_fr(); to29_261(proto.ark[r][j], &ark29[27 * r + 9 * j]); }
    uint32_t* d_ark;
    CK(hipMalloc(&d_ark, ark29.size() * 4));
    CK(hipMemcpy(d_ark, ark29.data(), ark29.size() * 4, hipMemcpyHostToDevice));
    PoseidonParams P{d_ark, nullptr, FULL, PART};

    // ---- 1. correctness: 200 proofs, k = 24 elements, ragged trailing zeros
    {
        const uint32_t B = 200, K = 24;
        std::vector<uint8_t> dig(32 * B);
        std::vector<Fr> el((size_t)B * K), el2((size_t)B * K);
        for (auto& x : dig) x = (uint8_t)next64();
        for (auto& x : el) x = rand_fr();
        for (auto& x : el2) x = rand_fr();
        for (uint32_t b = 0; b < B; b++) {   // trailing zeros: none, some, all
            const uint32_t z = (b % 7 == 3) ? K : (b % 5 == 1 ? b % K : 0);
            for (uint32_t i = K - z; i < K; i++) el2[(size_t)b * K + i] = Fr{{0, 0, 0, 0}};
        }
        uint32_t *d_state, *d_seeds, *d_lens; uint8_t* d_dig; fr *d_el, *d_el2;
        CK(hipMalloc(&d_state, B * kSpongeWords * 4)); CK(hipMalloc(&d_seeds, 3 * B * 32)); CK(hipMalloc(&d_lens, B * 4));
        CK(hipMalloc(&d_dig, dig.size())); CK(hipMalloc(&d_el, el.size() * 32)); CK(hipMalloc(&d_el2, el2.size() * 32));
        CK(hipMemcpy(d_dig, dig.data(), dig.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_el, el.data(), el.size() * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_el2, el2.data(), el2.size() * 32, hipMemcpyHostToDevice));
        std::vector<uint32_t> seeds(3 * B * 8), lens(B);
        std::vector<std::array<uint8_t, 32>> want[4];
        std::vector<HostSponge> hs(B, proto);
        int bad = 0;
        auto check = [&](int nsq, const char* what) {
            hipMemcpy(seeds.data(), d_seeds, (size_t)nsq * B * 32, hipMemcpyDeviceToHost);
            for (int j = 0; j < nsq; j++)
                for (uint32_t b = 0; b < B; b++) {
                    uint8_t w[32];
                    hs[b].squeeze_seed(w);
                    if (memcmp(w, &seeds[((size_t)j * B + b) * 8], 32) != 0) { if (bad < 5) printf("MISMATCH %s squeeze %d proof %u\n", what, j, b); bad++; }
                }
        };
        SpongeArgs a{};
        a.state = d_state; a.P = P; a.batch = B; a.seeds = d_seeds;
        // absorb(root), squeeze
        a.kind = kAbsorbDigest; a.digests = d_dig; a.digest_stride = 32; a.nsqueeze = 1; a.reset = 1;
        LAUNCH_SPONGE(B, 0, a);
        CK(hipDeviceSynchronize());
        for (uint32_t b = 0; b < B; b++) hs[b].absorb_bytes(&dig[32 * b], 32);
        check(1, "digest");
        // absorb(K elements), squeeze twice
        a.kind = kAbsorbElems; a.src = d_el; a.src_proof = K; a.count = K; a.trim = 0; a.nsqueeze = 2; a.reset = 0;
        LAUNCH_SPONGE(B, 0, a);
        CK(hipDeviceSynchronize());
        for (uint32_t b = 0; b < B; b++) hs[b].absorb(std::vector<Fr>(el.begin() + (size_t)b * K, el.begin() + (size_t)(b + 1) * K));
        check(2, "elements");
        // absorb(trimmed polynomial), squeeze three times (lanes now differ in length, mode and position)
        a.src = d_el2; a.trim = 1; a.lens_out = d_lens; a.nsqueeze = 3;
        LAUNCH_SPONGE(B, 0, a);
        CK(hipDeviceSynchronize());
        hipMemcpy(lens.data(), d_lens, B * 4, hipMemcpyDeviceToHost);
        for (uint32_t b = 0; b < B; b++) {
            std::vector<Fr> v(el2.begin() + (size_t)b * K, el2.begin() + (size_t)(b + 1) * K);
            while (!v.empty() && !(v.back().l[0] | v.back().l[1] | v.back().l[2] | v.back().l[3])) v.pop_back();
            if (lens[b] != v.size()) { if (bad < 5) printf("MISMATCH trimmed length proof %u: %u vs %zu\n", b, lens[b], v.size()); bad++; }
            hs[b].absorb(v);
        }
        check(3, "trimmed");
        // one more absorb of ONE element then an odd squeeze position is not reachable through squeeze_bytes(32); absorb 3 (odd) elements
        a.src = d_el; a.src_proof = K; a.count = 3; a.trim = 0; a.lens_out = nullptr; a.nsqueeze = 1;
        LAUNCH_SPONGE(B, 0, a);
        CK(hipDeviceSynchronize());
        for (uint32_t b = 0; b < B; b++) hs[b].absorb(std::vector<Fr>(el.begin() + (size_t)b * K, el.begin() + (size_t)b * K + 3));
        check(1, "odd count");
        // distinct indices from the last seeds: (n, t) = (1024, 156), (32, 32), (64, 40: complement), (65536, 156)
        const uint32_t shapes[4][2] = {{1024, 156}, {32, 32}, {64, 40}, {65536, 156}};
        for (auto& sh : shapes) {
            const uint32_t n = sh[0], t = sh[1], words = n >= 32 ? n / 32 : 1;
            uint32_t *d_bm, *d_idx;
            CK(hipMalloc(&d_bm, (size_t)B * words * 4)); CK(hipMalloc(&d_idx, (size_t)B * t * 4));
            IndexArgs ia{d_seeds, d_bm, d_idx, B, n, t};
            hipLaunchKernelGGL(distinct_indices_kernel, dim3((B + 63) / 64), dim3(64), 0, 0, ia);
            CK(hipDeviceSynchronize());
            std::vector<uint32_t> got((size_t)B * t);
            hipMemcpy(got.data(), d_idx, got.size() * 4, hipMemcpyDeviceToHost);
            for (uint32_t b = 0; b < B; b++) {
                const std::vector<uint32_t> w = host_indices(n, t, &seeds[(size_t)b * 8]);
                if (w.size() != t || memcmp(w.data(), &got[(size_t)b * t], t * 4) != 0) { if (bad < 5) printf("MISMATCH indices n=%u t=%u proof %u\n", n, t, b); bad++; }
            }
            hipFree(d_bm); hipFree(d_idx);
        }
        printf("correctness: %s (%d mismatches; 200 proofs x {digest, 24 elements, trimmed, odd count} + 4 index shapes)\n", bad ? "FAILED" : "ok", bad);
        if (bad) return 2;
    }

    // ---- 2. timing: absorb 256 elements = 128 permutations per lane
    const uint32_t K = 256, PERMS = 128;
    const uint32_t waves_list[] = {1, 4, 16, 64, 256, 1024, 4096};
    uint32_t* d_sink; CK(hipMalloc(&d_sink, 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double us_alone_16 = 0, us_loaded_16 = 0;
    for (int loaded = 0; loaded < 2; loaded++) {
        printf("%s\n  waves  proofs  ms/launch  us/permutation  permutation-lanes/s\n", loaded ? "beside a VALU-bound kernel (mul29 chains, 4096 workgroups x 256):" : "alone:");
        for (uint32_t waves : waves_list) {
            if (loaded && waves > 256) continue;
            const uint32_t B = 64 * waves;
            uint32_t* d_state; fr* d_el;
            CK(hipMalloc(&d_state, (size_t)B * kSpongeWords * 4)); CK(hipMalloc(&d_el, (size_t)B * K * 32));
            CK(hipMemset(d_el, 1, (size_t)B * K * 32));   // 0x0101.. < p
            SpongeArgs a{};
            a.state = d_state; a.P = P; a.batch = B; a.kind = kAbsorbElems; a.src = d_el; a.src_proof = K; a.count = K; a.reset = 1;
            LAUNCH_SPONGE(B, s1, a);   // warm-up
            CK(hipStreamSynchronize(s1));
            if (loaded) hipLaunchKernelGGL(valu_load_kernel, dim3(4096), dim3(256), 0, s2, d_sink, 60000u / 16);
            CK(hipEventRecord(e0, s1));
            LAUNCH_SPONGE(B, s1, a);
            CK(hipEventRecord(e1, s1));
            CK(hipStreamSynchronize(s1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipStreamSynchronize(s2));
            const double us = 1e3 * ms / PERMS;
            printf("  %5u  %6u  %9.3f  %14.2f  %.3e\n", waves, B, ms, us, (double)B * PERMS / (ms * 1e-3));
            if (waves == 16) (loaded ? us_loaded_16 : us_alone_16) = us;
            hipFree(d_state); hipFree(d_el);
        }
    }
    // ---- 2b. TWO sponge kernels at once (two provers in flight), 4 waves each: on plain streams the dispatcher puts the first
    // workgroups of every kernel on the same CUs -- the waves share SIMDs and each chain runs at half speed; streams created
    // with disjoint CU masks (hipExtStreamCreateWithCUMask) keep them apart
    {
        const uint32_t waves = 4, B = 64 * waves, NK = 4;
        uint32_t* d_state[NK]; fr* d_el[NK];
        for (uint32_t i = 0; i < NK; i++) {
            CK(hipMalloc(&d_state[i], (size_t)B * kSpongeWords * 4)); CK(hipMalloc(&d_el[i], (size_t)B * K * 32));
            CK(hipMemset(d_el[i], 1, (size_t)B * K * 32));
        }
        for (int masked = 0; masked < 3; masked++) {
            hipStream_t st[NK];
            for (uint32_t i = 0; i < NK; i++) {
                if (masked) {
                    // 256 CUs = 8 words; slot i takes 8 CUs: masked == 1: bits [8 i, 8 i + 8) of word 0..; masked == 2: one bit per word
                    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (masked == 1) mask[(8 * i) / 32] = 0xffu << ((8 * i) % 32);
                    else for (int w = 0; w < 8; w++) mask[w] = 1u << (4 + i);
                    hipError_t e = hipExtStreamCreateWithCUMask(&st[i], 8, mask);
                    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 3; }
                } else {
                    CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
                }
            }
            for (uint32_t nk = 1; nk <= NK; nk *= 2) {
                hipEvent_t b0[NK], b1[NK];
                for (uint32_t i = 0; i < nk; i++) { CK(hipEventCreate(&b0[i])); CK(hipEventCreate(&b1[i])); }
                for (int rep = 0; rep < 2; rep++) {
                    for (uint32_t i = 0; i < nk; i++) {
                        SpongeArgs a{};
                        a.state = d_state[i]; a.P = P; a.batch = B; a.kind = kAbsorbElems; a.src = d_el[i]; a.src_proof = K; a.count = K; a.reset = 1;
                        CK(hipEventRecord(b0[i], st[i]));
                        hipLaunchKernelGGL(sponge_kernel<true>, dim3(waves), dim3(64), 0, st[i], a);
                        CK(hipEventRecord(b1[i], st[i]));
                    }
                    for (uint32_t i = 0; i < nk; i++) CK(hipStreamSynchronize(st[i]));
                }
                float worst = 0;
                for (uint32_t i = 0; i < nk; i++) { float ms = 0; CK(hipEventElapsedTime(&ms, b0[i], b1[i])); worst = std::max(worst, ms); }
                printf("  %u sponge kernels of %u waves at once on %s streams: %.2f ms each (%.1f us per permutation)\n", nk, waves,
                       masked == 0 ? "plain" : (masked == 1 ? "CU-masked (8 adjacent bits)" : "CU-masked (one bit per mask word)"), worst, 1e3 * worst / PERMS);
            }
            for (uint32_t i = 0; i < NK; i++) CK(hipStreamDestroy(st[i]));
        }
    }
    // ---- 3. the projection: a Poseidon-shape proof is 326 permutations; the device's other work for a batch
    // of 64 proofs is 3.6 ms.  Two chains of 512 proofs in flight (1024 in all): each chain costs
    // 8 x 3.6 ms of device work + the sponge's latency, the two overlap.
    const double lat_ms = 326 * us_loaded_16 * 1e-3, lat_alone = 326 * us_alone_16 * 1e-3, dev512 = 8 * 3.6;
    printf("sponge latency per proof chain: %.1f ms alone, %.1f ms beside VALU-bound work\n", lat_alone, lat_ms);
    printf("projection, 2 chains x 512 proofs in flight: %.0f proofs/s (one chain of 1024: %.0f); kill criterion: < 12000\n",
           1024.0 / (std::max(2 * dev512, dev512 + lat_ms) * 1e-3), 1024.0 / ((2 * dev512 + lat_ms) * 1e-3));
    return 0;
}
