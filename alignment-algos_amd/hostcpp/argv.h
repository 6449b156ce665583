// argv.h — command line: "--KEY value" pairs go to the ParamStore, everything else is positional / switches.
// Mirrors Argv (reference argv.h:24-59, argv.cpp:46-108).  Own implementation.
#ifndef ALN_HOST_ARGV_H
#define ALN_HOST_ARGV_H
#include <string>
#include <vector>
#include "pstore.h"
// standard headers the reference's argv.h hands on to its includers
#include <sstream>
using namespace std;   // as the reference's argv.h does at header scope: sources written against it name string, vector, cerr ... unqualified

class Argv : public ParamStore {
 public:
  Argv(int argc, const char** argv) : help_(false) {
    for (int i = 1; i < argc; ++i) {
      std::string s(argv[i]);
      if (s == "-help") help_ = true;
    }
    for (int i = 1; i < argc; ++i) {
      std::string s(argv[i]);
      if (s.compare(0, 2, "--") == 0) {
        if (i + 1 >= argc) throw std::string("Argument missing for ") + s;
        setValue(s.substr(2), argv[++i]);
      } else {
        rest_.push_back(s);
      }
    }
  }
  int count() { return (int)rest_.size(); }
  bool help() { return help_; }
  std::stringstream& getArg(int c, bool cleanbuff = true, bool eraseafter = false) {
    if (cleanbuff) { abuf_.str(""); abuf_.clear(); }
    if (c < 0 || c >= (int)rest_.size()) throw std::string("Argument missing");
    abuf_ << rest_[c];
    if (eraseafter) rest_.erase(rest_.begin() + c);
    return abuf_;
  }
  bool getSwitch(const char* sw, bool eraseafter = true) {
    for (size_t i = 0; i < rest_.size(); ++i)
      if (rest_[i] == sw) { if (eraseafter) rest_.erase(rest_.begin() + i); return true; }
    return false;
  }
  // the c words following switch `sw`
  std::stringstream& getSwitch(const char* sw, int c, bool cleanbuff = true, bool eraseafter = true) {
    if (cleanbuff) { abuf_.str(""); abuf_.clear(); }
    size_t at = rest_.size();
    for (size_t i = 0; i < rest_.size(); ++i) if (rest_[i] == sw) { at = i; break; }
    if (at == rest_.size()) { if (c > 0) throw std::string("Switch arg missing for ") + sw; return abuf_; }
    if (at + (size_t)c >= rest_.size() && c > 0) throw std::string("Switch arg missing for ") + sw;
    for (int k = 1; k <= c; ++k) abuf_ << rest_[at + k] << " ";
    if (eraseafter) rest_.erase(rest_.begin() + at, rest_.begin() + at + c + 1);
    return abuf_;
  }
 private:
  bool help_;
  std::vector<std::string> rest_;
  std::stringstream abuf_;
};

template <class param_t>
Argv& operator>>(Argv& a, param_t& p) { p.read(&a); return a; }
#endif
