// alib.h — alignment parameters.  Same names, values and keys as the reference (alib.h:20-46, alib.cpp:16-47).
#ifndef ALN_HOST_ALIB_H
#define ALN_HOST_ALIB_H
#include <string>
#include "pstore.h"

enum align_t {            // alignment overhang treatment; numeric values are part of the ALIGN_MODE flag
  global_local = 0,       // overhangs penalised in the template, free in the query
  global = 1,             // overhangs penalised
  local_global = 2,       // overhangs penalised in the query, free in the template
  local = 3,              // local alignment (scores clipped at 0)
  semi_local = 4          // overhangs free
};

class AliParams {
 public:
  AliParams() : align_type(semi_local), gap_init_penalty(4.73f), gap_extn_penalty(0.34f) {}
  void read(ParamStore* p) {
    std::string s;
    s = "ALIGN_MODE";
    if (p->find(s)) { int v = align_type; p->getValue(s) >> v; align_type = static_cast<align_t>(v); }
    s = "GAP_INIT_PENALTY";
    if (p->find(s)) p->getValue(s) >> gap_init_penalty;
    s = "GAP_EXTN_PENALTY";
    if (p->find(s)) p->getValue(s) >> gap_extn_penalty;
    s = "SUB_MATRIX";
    if (p->find(s)) p->getValue(s) >> submatrix_fn;
  }
  align_t align_type;
  float gap_init_penalty;
  float gap_extn_penalty;
  std::string submatrix_fn;
};
#endif
