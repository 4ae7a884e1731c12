// tests/cpp/jit_wrap_check.cpp -- jit_module.h: the code object jit_wrap makes by growing the template's .text must be,
// byte for byte, the one the assembler and linker make from the same (page-padded) code.  No device involved.  Test code.
#include <cstdio>
#include <cstring>
#include <vector>

#include "escoin_plan.h"
#include "jit_module.h"

using namespace escoin;

int main() {
  int bad = 0;
  for (size_t bytes : {(size_t)4096, (size_t)8192, (size_t)4100, (size_t)64, (size_t)200000, (size_t)3000004}) {
    std::vector<uint32_t> code(bytes / 4);
    for (size_t i = 0; i < code.size(); ++i) code[i] = 0xBF800000u | (uint32_t)((i * 2654435761u) & 0xFFFF);
    std::vector<uint32_t> padded = code;
    while ((padded.size() * 4) % 4096) padded.push_back(0xBF800000u);
    std::vector<char> a, w;
    const int ra = jit_assemble(padded, &a), rw = jit_wrap(code, &w);
    const bool same = ra == 0 && rw == 0 && a.size() == w.size() && std::memcmp(a.data(), w.data(), a.size()) == 0;
    printf("%zu bytes of code: assembled %zu B (rc %d), wrapped %zu B (rc %d): %s\n", bytes, a.size(), ra, w.size(), rw, same ? "identical" : "DIFFERENT");
    bad += !same;
  }
  printf(bad ? "FAILED\n" : "all cases OK\n");
  return bad;
}
