// pstore.h — key:value parameter store (host plumbing for the drivers).
// Mirrors the reference's ParamStore surface (pstore.h / pstore.cpp:40-77): find(), getValue() as a stream,
// setValue(), and "KEY: value" lines with '#' comments.  Own implementation.
#ifndef ALN_HOST_PSTORE_H
#define ALN_HOST_PSTORE_H
#include <istream>
#include <map>
#include <sstream>
#include <string>
using namespace std;   // as the reference's pstore.h does at header scope: sources written against it name string, vector, cerr ... unqualified

class ParamStore {
 public:
  virtual ~ParamStore() {}
  bool find(const std::string& key) const { return values_.count(key) != 0; }
  // a fresh stream positioned at the start of the value: `p->getValue(s) >> x`
  std::stringstream& getValue(const std::string& key) {
    buf_.str(values_[key]);
    buf_.clear();
    return buf_;
  }
  bool setValue(const std::string& key, const std::string& value) { values_[key] = value; return true; }
  // one "KEY: value" line; blank lines and '#' comments are skipped.  false at end of input.
  bool extract(std::istream& in, std::string& key, std::string& value) {
    std::string line;
    do {
      if (!std::getline(in, line)) return false;
    } while (line.empty() || line[0] == '#');
    std::string::size_type colon = line.find(':');
    if (colon == std::string::npos) throw std::string("Param parse error");
    key = line.substr(0, colon);
    std::string::size_type v = line.find_first_not_of(" \t", colon + 1);
    value = (v == std::string::npos) ? std::string() : line.substr(v);
    return true;
  }
 protected:
  std::map<std::string, std::string> values_;
  std::stringstream buf_;
};
#endif
