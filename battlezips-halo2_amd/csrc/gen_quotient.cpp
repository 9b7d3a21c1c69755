// Build-time generator of the quotient kernels of the reference's two circuits (run by the Makefile, host only, no GPU):
//   ShotCircuit (src/circuits/shot.rs, 24 gates / 81 constraint polynomials) and BoardCircuit (src/circuits/board.rs,
//   57 gates / 140 polynomials) -> their VM v2 programs as straight-line HIP (bzh_quotient_source_for_circuit), one
//   namespace + kernel + launcher each, and the table bzh_builtin_quotients() that bzh_pk_create searches by program hash.
// The program depends on the constraint system only -- not on k, the SRS or a witness -- so one kernel per circuit serves
// every table size; the circuits are built here at their smallest k.
// Usage: gen_quotient > quotient_builtin.hip      (linked against the library's objects, minus quotient_builtin.o)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/bzh2.h"

int main() {
    struct Item {
        int kind;
        unsigned k;
        const char* name;
    };
    const Item items[] = {{BZH_CIRCUIT_SHOT, 11, "ShotCircuit"}, {BZH_CIRCUIT_BOARD, 12, "BoardCircuit"}};
    std::string body;
    std::vector<std::pair<unsigned long long, std::string>> table;
    for (const Item& it : items) {
        bzh_circuit* c = nullptr;
        int rc = bzh_circuit_create(it.kind, it.k, 0, &c);
        if (rc) {
            fprintf(stderr, "gen_quotient: bzh_circuit_create(%s) failed: %d\n", it.name, rc);
            return 1;
        }
        size_t blen = 0;
        bzh_circuit_blob(c, nullptr, 0, &blen);
        std::vector<uint8_t> blob(blen);
        rc = bzh_circuit_blob(c, blob.data(), blob.size(), &blen);
        bzh_circuit_free(c);
        if (rc) {
            fprintf(stderr, "gen_quotient: bzh_circuit_blob(%s) failed: %d\n", it.name, rc);
            return 1;
        }
        size_t slen = 0;
        uint64_t hash = 0;
        rc = bzh_quotient_source_for_circuit(BZH_CURVE_VESTA, blob.data(), blen, nullptr, 0, &slen, &hash);
        if (rc) {
            fprintf(stderr, "gen_quotient: no VM v2 program for %s: %d\n", it.name, rc);
            return 1;
        }
        std::string src(slen + 1, '\0');
        rc = bzh_quotient_source_for_circuit(BZH_CURVE_VESTA, blob.data(), blen, &src[0], src.size(), &slen, &hash);
        if (rc) return 1;
        src.resize(slen);
        bool seen = false;
        for (auto& t : table) seen |= t.first == (unsigned long long)hash;
        if (!seen) {
            body += "// ---- " + std::string(it.name) + " ----\n" + src + "\n";
            table.push_back({(unsigned long long)hash, it.name});
        }
    }
    printf("// GENERATED at build time by csrc/gen_quotient.cpp -- do not edit, do not commit.\n"
           "#include <hip/hip_runtime.h>\n#include \"field.cuh\"\n#include \"fe29.cuh\"\n#include \"../../include/bzh2.h\"\n\n%s", body.c_str());
    printf("static const bzh_builtin_quotient kTable[] = {\n");
    for (auto& t : table) printf("    {0x%016llxull, bzh_q_%016llx::launch, \"%s\", bzh_q29_%016llx::launch},\n", t.first, t.first, t.second.c_str(), t.first);
    printf("};\nextern \"C\" const bzh_builtin_quotient* bzh_builtin_quotients(size_t* count) {\n"
           "    if (count) *count = sizeof(kTable) / sizeof(kTable[0]);\n    return kTable;\n}\n");
    return 0;
}
