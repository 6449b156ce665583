// formats.h — stream manipulator tags: `in >> Formats::FastaIn() >> seq`, `out << Formats::FastaOut(60) << set`,
// `out << Formats::PIROut(60) << set`
// (reference formats.h:12-48).
#ifndef ALN_HOST_FORMATS_H
#define ALN_HOST_FORMATS_H
#include <string>
#include "alignment.h"
#include "gstrings.h"
// standard headers the reference's formats.h hands on to its includers
#include <iostream>
#include <sstream>

struct Formats {
  struct FastaOut { FastaOut(int len = 60) : line_length(len) {} int line_length; };
  struct PIROut { PIROut(int len = 60) : line_length(len) {} int line_length; };
  struct HMAPOut { HMAPOut(const char* sm = "", int len = 60) : line_length(len), submatrix(sm) {} int line_length; std::string submatrix; };
  struct FastaIn {
    FastaIn(const char* cs = "", bool flag = true) : head_tail(flag), find_me(cs) {}
    bool head_tail;
    std::string find_me;
  };
};
#endif
