// Minimal C++ caller of the host mirror: commits a 4m x k matrix and opens two columns.
// Built by `make -C ligero_amd/host` as a compile/link check of ligero.hpp against the C ABI;
// run it on a GPU box: ./example_commit
#include <cstdio>

#include "ligero.hpp"

int main() {
    try {
        const size_t m = 4, k = 4;
        ligero::LigeroCircuit lc(m, k);
        std::vector<std::vector<ligero::Fr>> rows(4 * m, std::vector<ligero::Fr>(k, ligero::Fr{{0, 0, 0, 0}}));
        rows[0][0] = ligero::Fr{{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};  // 1
        ligero::Commitment c = lc.commit(ligero::DenseMatrix(rows));
        auto opened = lc.open_columns({0, 31});
        std::printf("u_root = ");
        for (uint8_t b : c.u_root) std::printf("%02x", b);
        std::printf("\nopened %zu columns, path length %zu\n", opened.first.size(), opened.second[0].auth_path.size());
        return 0;
    } catch (const ligero::Error& e) {
        std::fprintf(stderr, "ligero error %d: %s\n", e.status, e.what());
        return 1;
    }
}
