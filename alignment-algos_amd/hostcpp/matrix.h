// matrix.h — dense row-major matrix with (r,c) and [r][c] access (reference matrix.h:146-233 uses valarray slices).
#ifndef ALN_HOST_MATRIX_H
#define ALN_HOST_MATRIX_H
#include <vector>
// standard headers the reference's matrix.h hands on to its includers
#include <valarray>
using namespace std;   // as the reference's matrix.h does at header scope: sources written against it name string, vector, cerr ... unqualified

template <class val_t>
class matrix {
 public:
  matrix(int nr, int nc) : nrows(nr), ncols(nc), v_((size_t)nr * nc) {}
  int size() const { return nrows * ncols; }
  int rows() const { return nrows; }
  int cols() const { return ncols; }
  val_t operator()(int r, int c) const { return v_[(size_t)r * ncols + c]; }
  val_t& operator()(int r, int c) { return v_[(size_t)r * ncols + c]; }
  val_t* operator[](int r) { return &v_[(size_t)r * ncols]; }
  const val_t* operator[](int r) const { return &v_[(size_t)r * ncols]; }
  val_t* data() { return v_.data(); }
  const val_t* data() const { return v_.data(); }
 private:
  int nrows, ncols;
  std::vector<val_t> v_;
};
#endif
