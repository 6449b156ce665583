// aaa_hip.cpp — the `aaa` driver (reference aa_ali.cpp:24-117, make target `aaa`) on the MI355X engine:
// BASELINE.json config 1.  Same flags, same stdout (score matrix, FASTA alignment block, timing lines); written
// against the same class names, so its body reads like the reference driver.  Differences, all documented in
// template is the FIRST FASTA record (B2, reproduced), SuboptFlags(bool,len) is called in the right
// argument order (B3), the FASTA reader does not duplicate a trailing line (B1).
#include <ctime>
#include <fstream>
#include <iostream>

#include "aa_seq.h"
#include "aasubalib.h"
#include "application.h"
#include "argv.h"
#include "cw.h"
#include "dpmatrix.h"
#include "fastaio.h"
#include "formats.h"
#include "noalib.h"
#include "optimal.h"
#include "pirio.h"
#include "rcfile.h"
#include "sequence.h"
#include "sflags.h"

using namespace std;

typedef AASubstitutionEval<AASequence, AASequence> AAEval;

static void usage() {
  cerr << endl << "Usage: aaa_hip [-opt] [-top file] [--KEY value ...] fasta_seqs" << endl << endl;
  cerr << "   Optimal and near-optimal alignments of the two sequences of a FASTA file (template first)," << endl;
  cerr << "   scored with a substitution matrix, computed on an MI355X." << endl;
  cerr << "   -opt           only the optimal alignment" << endl;
  cerr << "   -top <file>    parameter file (KEY: value lines)" << endl;
  cerr << "   --SUB_MATRIX f --ALIGN_MODE n --GAP_INIT_PENALTY x --GAP_EXTN_PENALTY x --NUM_SUBOPT n --DELTA_RATIO x" << endl << endl;
  exit(0);
}

namespace {

struct Settings {
  AliParams ali;
  ApplicationParams app;
  NOaliParams noa;
  bool optimal_only;
  string fasta;
};

// defaults <- ~/.hmaprc <- -top file <- --KEY value overrides, in the reference driver's order (aa_ali.cpp:33-58)
Settings read_settings(int argc, const char** argv) {
  if (argc == 0) usage();
  Argv args(argc, argv);
  if (args.help()) usage();
  Settings s;
  string topfile;
  if (args.getSwitch("-top", false)) args.getSwitch("-top", 1) >> topfile;
  s.optimal_only = args.getSwitch("-opt", true);
  RCfile rc;
  rc >> s.ali >> s.app >> s.noa;
  if (!topfile.empty()) {
    RCfile top(topfile);
    top >> s.ali >> s.app;
  }
  args >> s.ali >> s.app >> s.noa;
  if (args.count() != 1) usage();
  s.fasta = args.getArg(0).str();
  return s;
}

template <class Set>
void print_alignments(const Settings& s, Set& alignments) {
  if (s.app.output_format == oFASTA) cout << Formats::FastaOut(s.app.line_length) << alignments;
  else if (s.app.output_format == oPIR) cout << Formats::PIROut(s.app.line_length) << alignments;
  else { cerr << "Cannot use this format!\n"; exit(-1); }
}

}  // namespace

int main(int argc, const char** argv) {
  try {
    const clock_t started = clock();
    Settings s = read_settings(argc, argv);

    // the FIRST record is the template, the messages are the reference's
    AASequence templ, query;
    ifstream in(s.fasta.c_str());
    cerr << "Reading in query profile" << endl;
    in >> Formats::FastaIn() >> templ;
    cerr << "Reading in template profile" << endl;
    in >> Formats::FastaIn() >> query;

    BlosumMatrix blosum(s.ali.submatrix_fn.c_str());
    AAEval scoring(s.ali, blosum);
    DPMatrix<AASequence, AASequence, AAEval> dpm(query, templ, scoring, fwd, s.ali.align_type);
    cout << dpm << endl;

    const clock_t aligned_from = clock();
    Optimal<AASequence, AASequence, AAEval> best(s.ali.align_type);
    AlignmentSet<AASequence, AASequence, AAEval> alignments(dpm, best);
    if (!s.optimal_only) {
      SuboptFlags everywhere(true, templ.size());
      ConstrainedNearOptimal<AASequence, AASequence, AAEval> near_optimal(s.noa, everywhere);
      near_optimal.enumerate(dpm, alignments);
    }
    alignments.assignIdentity();
    const clock_t finished = clock();

    print_alignments(s, alignments);
    cout << "time for alignment was (sec) " << (finished - aligned_from) / (double)CLOCKS_PER_SEC << endl;
    cout << "total cpu time was (sec) " << (finished - started) / (double)CLOCKS_PER_SEC << endl << endl;
  } catch (string e) {
    cerr << e << endl;
    exit(-1);
  }
  return 0;
}
