// rcfile.h — defaults file ("~/.hmaprc" or a named file) of "KEY: value" lines.  Mirrors RCfile
// (reference rcfile.h, rcfile.cpp:16-50): a missing default file is only a warning.  Own implementation.
#ifndef ALN_HOST_RCFILE_H
#define ALN_HOST_RCFILE_H
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include "pstore.h"

class RCfile : public ParamStore {
 public:
  RCfile() : fname_("~/.hmaprc") {
    resolve(fname_);
    std::ifstream fin(fname_.c_str());
    if (!fin.good()) std::cerr << "No defaults file (~/.hmaprc).  Using programmed defaults." << std::endl;
    else load(fin);
  }
  explicit RCfile(const std::string& fn) : fname_(fn) {
    resolve(fname_);
    std::ifstream fin(fname_.c_str());
    if (!fin.good()) throw std::string("Cannot open parameter file ") + fname_;
    load(fin);
  }
 private:
  static void resolve(std::string& f) {
    if (!f.empty() && f[0] == '~') { const char* h = getenv("HOME"); f = std::string(h ? h : "") + f.substr(1); }
  }
  void load(std::istream& in) { std::string k, v; while (extract(in, k, v)) setValue(k, v); }
  std::string fname_;
};

template <class param_t>
RCfile& operator>>(RCfile& rc, param_t& p) { p.read(&rc); return rc; }
#endif
