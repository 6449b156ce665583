// submatrix.h — substitution matrices (reference submatrix.h:19-48, submatrix.cpp:16-54).
// BlosumMatrix reads the usual NCBI text: '#' comment lines, one line of column letters, then one row per letter
// ("label v v v ...").  Stored densely (the reference uses map<char,map<char,float>>); a letter outside the
// alphabet is reported instead of dereferencing end().
#ifndef ALN_HOST_SUBMATRIX_H
#define ALN_HOST_SUBMATRIX_H
#include <fstream>
#include <ostream>
#include <string>
#include <vector>
// standard headers the reference's submatrix.h hands on to its includers
#include <map>
using namespace std;   // as the reference's submatrix.h does at header scope: sources written against it name string, vector, cerr ... unqualified

class SubstitutionMatrix {
 public:
  bool hasLetter(char x) const { return alphabet.find(x) != std::string::npos; }
  float score(char a, char b) const {
    std::string::size_type i = alphabet.find(a), j = alphabet.find(b);
    if (i == std::string::npos || j == std::string::npos) throw std::string("Residue not in substitution matrix alphabet");
    return values_[i * alphabet.size() + j];
  }
  const std::string& letters() const { return alphabet; }
  const float* table() const { return values_.data(); }
  friend std::ostream& operator<<(std::ostream& os, SubstitutionMatrix& m) {
    for (size_t i = 0; i < m.alphabet.size(); ++i)
      for (size_t j = 0; j < m.alphabet.size(); ++j)
        os << m.alphabet[i] << m.alphabet[j] << ":" << m.values_[i * m.alphabet.size() + j] << std::endl;
    return os;
  }
 protected:
  std::string alphabet;
  std::vector<float> values_;
};

class BlosumMatrix : public SubstitutionMatrix {
 public:
  explicit BlosumMatrix(const char* filename) {
    std::ifstream in(filename);
    if (!in.good()) throw std::string("File not found (substitution matrix) ") + filename;
    std::string line;
    while (std::getline(in, line)) if (line.empty() || line[0] != '#') break;
    for (size_t k = 0; k < line.size(); ++k) if (line[k] != ' ' && line[k] != '\n' && line[k] != '\r' && line[k] != '\t') alphabet.push_back(line[k]);
    const size_t n = alphabet.size();
    values_.assign(n * n, 0.f);
    for (size_t i = 0; i < n; ++i) {
      std::string label;
      in >> label;
      for (size_t j = 0; j < n; ++j) in >> values_[i * n + j];
    }
  }
};
#endif
