// noalib.h — near-optimal alignment parameters (reference noalib.h:18-45, noalib.cpp:15-63): same fields, keys, defaults.
#ifndef ALN_HOST_NOALIB_H
#define ALN_HOST_NOALIB_H
#include <string>
#include "pstore.h"

class NOaliParams {
 public:
  NOaliParams()
      : number_suboptimal(200), subopt_per_round(200), delta_ratio(0.01f), k_limit(16), sort_limit(100),
        user_limit(100000), max_overlap(0.30f), final_overlap(0.30f), rounds(4) {}
  void read(ParamStore* p) {
    std::string s;
    s = "NUM_SUBOPT"; if (p->find(s)) p->getValue(s) >> number_suboptimal;
    s = "NUM_ROUND_SUBOPT"; if (p->find(s)) p->getValue(s) >> subopt_per_round;
    s = "DELTA_RATIO"; if (p->find(s)) p->getValue(s) >> delta_ratio;
    s = "K_LIMIT"; if (p->find(s)) p->getValue(s) >> k_limit;
    s = "USER_LIMIT"; if (p->find(s)) p->getValue(s) >> user_limit;
    s = "SORT_LIMIT"; if (p->find(s)) p->getValue(s) >> sort_limit;
    s = "MAX_OVERLAP"; if (p->find(s)) p->getValue(s) >> max_overlap;
    s = "FINAL_OVERLAP"; if (p->find(s)) p->getValue(s) >> final_overlap;
    s = "ROUNDS"; if (p->find(s)) p->getValue(s) >> rounds;
  }
  int number_suboptimal;
  int subopt_per_round;
  float delta_ratio;
  unsigned int k_limit;
  unsigned int sort_limit;
  unsigned int user_limit;
  float max_overlap;
  float final_overlap;
  unsigned int rounds;
};
#endif
