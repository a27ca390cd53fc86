// rc_tables.h -- genetic code and substitution matrices used by the scoring path
// (same information as the reference's src/code.c:26-88, own representation).
#pragma once
#include <cstdint>
#include <cstring>

namespace rc {

// amino-acid order of the BLOSUM triangles below
static const char kAaOrder[] = "ARNDCQEGHILKMFPSTWYV";

// standard genetic code; codon index = 16*n1 + 4*n2 + n3, A=0 C=1 G=2 T=3; '*' = stop
static const char kGeneticCode[] =
    "KNKNTTTTRSRSIIMI"
    "QHQHPPPPRRRRLLLL"
    "EDEDAAAAGGGGVVVV"
    "*Y*YSSSS*CWCLFLF";

// lower triangles (row-major, diagonal included) of the symmetric 20x20 matrices
static const int8_t kBlosum62Tri[210] = {
    4,
    -1, 5,
    -2, 0, 6,
    -2, -2, 1, 6,
    0, -3, -3, -3, 9,
    -1, 1, 0, 0, -3, 5,
    -1, 0, 0, 2, -4, 2, 5,
    0, -2, 0, -1, -3, -2, -2, 6,
    -2, 0, 1, -1, -3, 0, 0, -2, 8,
    -1, -3, -3, -3, -1, -3, -3, -4, -3, 4,
    -1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4,
    -1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5,
    -1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5,
    -2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6,
    -1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7,
    1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4,
    0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5,
    -3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11,
    -2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7,
    0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4};

static const int8_t kBlosum90Tri[210] = {
    5,
    -2, 6,
    -2, -1, 7,
    -3, -3, 1, 7,
    -1, -5, -4, -5, 9,
    -1, 1, 0, -1, -4, 7,
    -1, -1, -1, 1, -6, 2, 6,
    0, -3, -1, -2, -4, -3, -3, 6,
    -2, 0, 0, -2, -5, 1, -1, -3, 8,
    -2, -4, -4, -5, -2, -4, -4, -5, -4, 5,
    -2, -3, -4, -5, -2, -3, -4, -5, -4, 1, 5,
    -1, 2, 0, -1, -4, 1, 0, -2, -1, -4, -3, 6,
    -2, -2, -3, -4, -2, 0, -3, -4, -3, 1, 2, -2, 7,
    -3, -4, -4, -5, -3, -4, -5, -5, -2, -1, 0, -4, -1, 7,
    -1, -3, -3, -3, -4, -2, -2, -3, -3, -4, -4, -2, -3, -4, 8,
    1, -1, 0, -1, -2, -1, -1, -1, -2, -3, -3, -1, -2, -3, -2, 5,
    0, -2, 0, -2, -2, -1, -1, -3, -2, -1, -2, -1, -1, -3, -2, 1, 6,
    -4, -4, -5, -6, -4, -3, -5, -4, -3, -4, -3, -5, -2, 0, -5, -4, -4, 11,
    -3, -3, -3, -4, -4, -3, -4, -5, 1, -2, -2, -3, -2, 3, -4, -3, -2, 2, 8,
    -1, -3, -4, -5, -2, -3, -3, -5, -4, 3, 0, -3, 0, -2, -3, -2, -1, -3, -3, 5};

struct CodeTables {
  int pep[64];         // amino-acid index per codon, -1 = stop
  int blosum[20][20];  // selected matrix
  CodeTables(int which) {
    for (int c = 0; c < 64; c++) {
      const char *p = std::strchr(kAaOrder, kGeneticCode[c]);
      pep[c] = (kGeneticCode[c] == '*' || !p) ? -1 : int(p - kAaOrder);
    }
    const int8_t *tri = (which == 90) ? kBlosum90Tri : kBlosum62Tri;
    for (int i = 0, t = 0; i < 20; i++)
      for (int j = 0; j <= i; j++, t++) blosum[i][j] = blosum[j][i] = tri[t];
  }
};

// nucleotide code of the reference's ntMap (score.c:41, RNAcode.c:94-98): A0 C1 G2 T/U3, all else 0
static inline int nt_code(unsigned char c) {
  switch (c) {
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 0;
  }
}

}  // namespace rc
