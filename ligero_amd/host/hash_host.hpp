// Host-side Blake2s-256 (RFC 7693), SHA-256 (FIPS 180-4) and the Merkle path check of the
// verifier (ark-crypto-primitives merkle_tree::Path::verify with the reference's
// TestMerkleTreeParams, src/ligero/types.rs:18-26; call site src/ligero/mod.rs:985-995).
// Product code: independent of oracle/.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

namespace ligero {

using Digest = std::array<uint8_t, 32>;

namespace detail {
inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline uint32_t load32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t load32be(const uint8_t* p) { return (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24); }
}  // namespace detail

// ---- Blake2s-256, unkeyed, no salt / personalisation (parameter block 0x01010020)
class Blake2s {
public:
    Blake2s() {
        static const uint32_t iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        memcpy(h_, iv, sizeof(h_));
        h_[0] ^= 0x01010020u;
    }
    void update(const uint8_t* data, size_t len) {
        while (len > 0) {
            if (fill_ == 64) {  // a full buffer is only compressed once more input is known to follow
                t_ += 64;
                compress(buf_, false);
                fill_ = 0;
            }
            const size_t take = (64 - fill_ < len) ? 64 - fill_ : len;
            memcpy(buf_ + fill_, data, take);
            fill_ += take;
            data += take;
            len -= take;
        }
    }
    Digest finalize() {
        t_ += fill_;
        memset(buf_ + fill_, 0, 64 - fill_);
        compress(buf_, true);
        Digest out;
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)h_[i];
            out[4 * i + 1] = (uint8_t)(h_[i] >> 8);
            out[4 * i + 2] = (uint8_t)(h_[i] >> 16);
            out[4 * i + 3] = (uint8_t)(h_[i] >> 24);
        }
        return out;
    }

private:
    void compress(const uint8_t* block, bool last) {
        static const uint32_t iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        static const uint8_t sigma[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++) m[i] = detail::load32le(block + 4 * i);
        for (int i = 0; i < 8; i++) { v[i] = h_[i]; v[8 + i] = iv[i]; }
        v[12] ^= (uint32_t)t_;
        v[13] ^= (uint32_t)(t_ >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] = v[a] + v[b] + x; v[d] = detail::rotr32(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = detail::rotr32(v[b] ^ v[c], 12);
            v[a] = v[a] + v[b] + y; v[d] = detail::rotr32(v[d] ^ v[a], 8);
            v[c] = v[c] + v[d];     v[b] = detail::rotr32(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const uint8_t* s = sigma[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[8 + i];
    }
    uint32_t h_[8];
    uint64_t t_ = 0;
    uint8_t buf_[64];
    size_t fill_ = 0;
};

// ---- SHA-256, one shot
inline Digest sha256(const uint8_t* data, size_t len) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
        0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
        0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
        0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
        0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::vector<uint8_t> msg(data, data + len);
    msg.push_back(0x80);
    while (msg.size() % 64 != 56) msg.push_back(0);
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 7; i >= 0; i--) msg.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < msg.size(); off += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = detail::load32be(&msg[off + 4 * i]);
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = detail::rotr32(w[i - 15], 7) ^ detail::rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = detail::rotr32(w[i - 2], 17) ^ detail::rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t S1 = detail::rotr32(e, 6) ^ detail::rotr32(e, 11) ^ detail::rotr32(e, 25);
            const uint32_t ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = detail::rotr32(a, 2) ^ detail::rotr32(a, 13) ^ detail::rotr32(a, 22);
            const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    Digest out;
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
    return out;
}

// ---- Merkle path of one opened column (ark-crypto-primitives merkle_tree::Path)
struct MerklePath {
    Digest leaf_sibling_hash;
    std::vector<Digest> auth_path;  // siblings of the ancestors, root side first (log2 n - 1 of them)
    uint64_t leaf_index = 0;
};

// Path::verify(leaf_hash_param, two_to_one_param, root, leaf) with LeafHash = identity,
// bottom level SHA-256(LE64(32) || L || LE64(32) || R), upper levels SHA-256(L || R).
inline bool merkle_path_verify(const MerklePath& p, const Digest& root, const Digest& leaf) {
    const Digest& left = (p.leaf_index & 1) ? p.leaf_sibling_hash : leaf;
    const Digest& right = (p.leaf_index & 1) ? leaf : p.leaf_sibling_hash;
    uint8_t buf[80];
    const uint8_t len32[8] = {32, 0, 0, 0, 0, 0, 0, 0};
    memcpy(buf, len32, 8);
    memcpy(buf + 8, left.data(), 32);
    memcpy(buf + 40, len32, 8);
    memcpy(buf + 48, right.data(), 32);
    Digest cur = sha256(buf, 80);
    uint64_t index = p.leaf_index >> 1;
    for (size_t level = p.auth_path.size(); level-- > 0;) {  // walk from the leaves up: last entry first
        const Digest& sib = p.auth_path[level];
        uint8_t b2[64];
        if (index & 1) {
            memcpy(b2, sib.data(), 32);
            memcpy(b2 + 32, cur.data(), 32);
        } else {
            memcpy(b2, cur.data(), 32);
            memcpy(b2 + 32, sib.data(), 32);
        }
        cur = sha256(b2, 64);
        index >>= 1;
    }
    return cur == root;
}

}  // namespace ligero
