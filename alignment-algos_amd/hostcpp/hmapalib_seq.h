// hmapalib_seq.h — HMAP profile sequences (reference hmapalib_seq.h:28-99, parser hmapalib_seq.cpp:68-117,182-243)
// without the Troll-only members (HM_Data rdata, struct.h).  Text format:
//   [PDB: ...]            optional
//   ID : name / DE : text / SR : text / EVD: a b / LEN: n
//   n records:  idx olc p1..p20 (percent)  - g0 g1 g2 g3 motif_v motif_c  * helix strand coil sse_conf surf_v surf_c
//   //
// Head '^' and tail '$' elements are added; their gap values copy the first / last residue (:237-238).
#ifndef ALN_HOST_HMAPALIB_SEQ_H
#define ALN_HOST_HMAPALIB_SEQ_H
#include <fstream>
#include <istream>
#include <string>
#include <valarray>
#include <vector>
#include "sequence.h"
#include "sflags.h"

class HMAPElem : public SequenceElem {
 public:
  std::valarray<float> aa_profile;
  std::valarray<float> gap_values;
  float motif_value, motif_confid;
  std::valarray<float> sse_values;
  float sse_confid, surfacc_value, surfacc_confid;
  unsigned int lods_type;
  float hydropathy;
  struct HM_Data { int isse; HM_Data() : isse(-1) {} } rdata;   // index of the SSE the residue belongs to (struct.h; -1 = none)
  HMAPElem() : aa_profile(20), gap_values(4), motif_value(0), motif_confid(0), sse_values(3), sse_confid(0), surfacc_value(0),
               surfacc_confid(0), lods_type(0), hydropathy(0) {}
  explicit HMAPElem(std::istream& in) : aa_profile(20), gap_values(4), sse_values(3) { readHMAP(in); }
  float gap_init() const { return gap_values[0]; }
  float gap_extn() const { return gap_values[1]; }
  void gap_init(float gi) { gap_values[0] = gi; }
  void gap_extn(float ge) { gap_values[1] = ge; }
  float p_helix() const { return sse_values[0]; }
  float p_strand() const { return sse_values[1]; }
  float p_coil() const { return sse_values[2]; }
  void readHMAP(std::istream& in) {
    int idx; char mark;
    in >> idx >> olc;
    for (int i = 0; i < 20; ++i) { in >> aa_profile[i]; aa_profile[i] /= 100.0f; }
    in >> mark;
    if (mark != '-') throw std::string("Parse error before '-'");
    for (int i = 0; i < 4; ++i) in >> gap_values[i];
    in >> motif_value >> motif_confid >> mark;
    if (mark != '*') throw std::string("Parse error before '*'");
    for (int i = 0; i < 3; ++i) in >> sse_values[i];
    in >> sse_confid >> surfacc_value >> surfacc_confid;
    unsigned int t = 3, c = 0;
    if (sse_values[0] > .5f) t = 0;
    if (sse_values[1] > .5f) t = 1;
    if (sse_values[2] > .5f) t = 2;
    if (sse_confid > .33f) c = 1;
    if (sse_confid > .66f) c = 2;
    lods_type = t * 3 + c;
    std::string rest;
    std::getline(in, rest);
  }
};

class HMAPSequence : public Sequence<HMAPElem*> {
 public:
  explicit HMAPSequence(const char* fn) : evd1_field(0), evd2_field(0) {
    std::ifstream in(fn);
    if (!in.good()) throw std::string("Error reading file");
    readHMAP(in);
  }
  explicit HMAPSequence(std::istream& in) : evd1_field(0), evd2_field(0) { readHMAP(in); }
  ~HMAPSequence() { for (size_t i = 0; i < size(); ++i) delete (*this)[i]; }
  std::string de_field, sr_field;
  float evd1_field, evd2_field;
  // one character per element: '^' / '$', H / E for a confident helix / strand (probability and confidence >= 0.5), h / e for a
  // weak one, ' ' otherwise (hmapalib_seq.cpp:245-270)
  const std::string* getSSEString() const {
    if (sse_string.empty()) {
      for (size_t i = 0; i < size(); ++i) {
        const HMAPElem* el = (*this)[i];
        char sse;
        const float helix = el->p_helix(), strand = el->p_strand(), coil = el->p_coil(), confid = el->sse_confid;
        if (el->isHead()) sse = SequenceElem::Head;
        else if (el->isTail()) sse = SequenceElem::Tail;
        else if (helix > strand && helix > coil) sse = (helix < .5 || confid < .5) ? 'h' : 'H';
        else if (strand > helix && strand > coil) sse = (strand < .5 || confid < .5) ? 'e' : 'E';
        else sse = ' ';
        sse_string.push_back(sse);
      }
    }
    return &sse_string;
  }
  // loop positions (p_coil > 0.3) are not suboptimal regions, everything else including the sentinels is (hmapalib_seq.cpp:272-282)
  void getDefaultFlags(SuboptFlags& sof) {
    sof.Set(0, true);
    for (unsigned int i = 1; i <= seq_length; ++i) sof.Set(i, !(at(i)->p_coil() > 0.3f));
    sof.Set(seq_length + 1, true);
  }
 protected:
  HMAPSequence() : evd1_field(0), evd2_field(0) {}
  static std::string field(std::istream& in) { std::string k; std::getline(in, k, ':'); return k; }
  void readHMAP(std::istream& in) {
    std::string rest, k = field(in);
    if (k == "PDB") { std::getline(in, rest); k = field(in); }
    if (k != "ID ") throw std::string("Parse error before 'ID'");
    in >> seq_name; std::getline(in, rest);
    if (field(in) != "DE ") throw std::string("Parse error before 'DE'");
    in >> de_field; std::getline(in, rest);
    if (field(in) != "SR ") throw std::string("Parse error before 'SR'");
    in >> sr_field; std::getline(in, rest);
    if (field(in) != "EVD") throw std::string("Parse error before 'EVD'");
    in >> evd1_field >> evd2_field; std::getline(in, rest);
    if (field(in) != "LEN") throw std::string("Parse error before 'LEN'");
    in >> seq_length; std::getline(in, rest);
    reserve(seq_length + 2);
    HMAPElem* head = new HMAPElem();
    head->olc = SequenceElem::Head; head->index = 0;
    push_back(head);
    for (unsigned int i = 0; i < seq_length; ++i) {
      HMAPElem* e = new HMAPElem(in);
      e->index = (int)i + 1;
      push_back(e);
    }
    HMAPElem* tail = new HMAPElem();
    tail->olc = SequenceElem::Tail; tail->index = (int)seq_length + 1;
    push_back(tail);
    if (seq_length > 0) { head->gap_values = at(1)->gap_values; tail->gap_values = at(seq_length)->gap_values; }
    std::getline(in, rest);
    if (rest != "//") throw std::string("end of profile '//' not found");
  }
 private:
  mutable std::string sse_string;
  HMAPSequence(const HMAPSequence&);
  HMAPSequence& operator=(const HMAPSequence&);
};

// Structure-annotated template (reference gn2lib_seq.h:32-57).  The reference fills the structural members from a PDB file
// through the Troll library, which does not exist here: the caller fills them (plain containers, same names and index
// conventions).  Hmap2Eval uses only the HMAP part; Gn2Eval (gn2_eval.h) reads all of them.
class SMAPSequence : public HMAPSequence {
 public:
  explicit SMAPSequence(const char* fn) : HMAPSequence(fn) {}
  explicit SMAPSequence(std::istream& in) : HMAPSequence(in) {}
  std::vector<std::vector<unsigned long> > brokenhb;     // [i-2][j], i = 2 .. seq_length+1, j < i-1   (gn2_eval.cpp:136-157)
  std::vector<std::vector<float> > distance;             // same triangular indexing
  std::vector<float> weighted_contact_number;            // one per element, sentinels included (seq_length + 2)
};
#endif
