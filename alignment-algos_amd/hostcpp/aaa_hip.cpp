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

int main(int argc, const char** argv) {
  try {
    clock_t t0 = clock();
    if (argc == 0) usage();
    Argv args(argc, argv);
    if (args.help()) usage();
    string topfile;
    if (args.getSwitch("-top", false)) args.getSwitch("-top", 1) >> topfile;
    bool optflag = args.getSwitch("-opt", true);

    AliParams ali_params;
    ApplicationParams app_params;
    NOaliParams noa_params;
    RCfile default_rc;
    default_rc >> ali_params >> app_params >> noa_params;
    if (!topfile.empty()) {
      RCfile top_rc(topfile);
      top_rc >> ali_params >> app_params;
    }
    args >> ali_params >> app_params >> noa_params;
    if (args.count() != 1) usage();

    AASequence query, templ;
    ifstream seqs(args.getArg(0).str().c_str());
    cerr << "Reading in query profile" << endl;
    seqs >> Formats::FastaIn() >> templ;
    cerr << "Reading in template profile" << endl;
    seqs >> Formats::FastaIn() >> query;

    BlosumMatrix blosum(ali_params.submatrix_fn.c_str());
    AAEval ge(ali_params, blosum);

    DPMatrix<AASequence, AASequence, AAEval> dpm(query, templ, ge, fwd, ali_params.align_type);
    cout << dpm << endl;

    clock_t t1 = clock();
    Optimal<AASequence, AASequence, AAEval> opt(ali_params.align_type);
    AlignmentSet<AASequence, AASequence, AAEval> alignments(dpm, opt);
    if (!optflag) {
      SuboptFlags subopt(true, templ.size());
      ConstrainedNearOptimal<AASequence, AASequence, AAEval> cno(noa_params, subopt);
      cno.enumerate(dpm, alignments);
    }
    alignments.assignIdentity();
    clock_t t2 = clock();

    switch (app_params.output_format) {
      case oFASTA:
        cout << Formats::FastaOut(app_params.line_length) << alignments;
        break;
      case oPIR:
        cout << Formats::PIROut(app_params.line_length) << alignments;
        break;
      default:
        cerr << "Cannot use this format!\n";
        exit(-1);
    }
    cout << "time for alignment was (sec) " << (t2 - t1) / (double)CLOCKS_PER_SEC << endl;
    cout << "total cpu time was (sec) " << (t2 - t0) / (double)CLOCKS_PER_SEC << endl << endl;
  } catch (string e) {
    cerr << e << endl;
    exit(-1);
  }
  return 0;
}
