// nalign2_hip.cpp — the profile-profile driver (reference nalign2.cpp:24-174, make target `gn2`/`nalign2`) on the MI355X
// engine: query HMAP profile x template profile, Hmap2Eval, global DPMatrix, Optimal, then ConstrainedNearOptimal over the
// template's default flags (loops excluded) or a flag file, or UnconstrainedNearOptimal (-ucw); FASTA or PIR output.
// Same flags and parameter handling as the reference driver.  Differences: the template is read as an HMAP profile (the
// reference's SMAPSequence also loads a PDB structure through the Troll library, which Hmap2Eval never looks at);
// -crcw (experimental, reads out of bounds in the reference: crcw.h:366-402) is refused; -kscw runs
// the engine's KSConstrainedNearOptimal (parity unpinned).  The reference binary cannot be built here, so this driver's stdout has no golden; its pieces
// (HMAP parser, Hmap2Eval, DPMatrix, Optimal, cw, writers) are each checked against the oracle / real reference.
#include <ctime>
#include <fstream>
#include <iostream>

#include "application.h"
#include "argv.h"
#include "cw.h"
#include "dpmatrix.h"
#include "fastaio.h"
#include "formats.h"
#include "hmap2_eval.h"
#include "hmapio.h"
#include "kscw.h"
#include "optimal.h"
#include "pirio.h"
#include "rcfile.h"
#include "sflags.h"
#include "ucw.h"

using namespace std;

static void usage() {
  cerr << endl << "Usage: nalign2_hip query.prof template.prof [template.flag]" << endl << endl;
  cerr << "   Optimal and near-optimal profile-profile alignments (Hmap2Eval) computed on an MI355X" << endl << endl;
  cerr << "   template.flag  specify regions for suboptimal alignment" << endl;
  cerr << "   -opt           just do an optimal alignment (-ucw & template.flag are ignored)" << endl;
  cerr << "   -ucw           do standard waterman suboptimal alignment (template.flag is ignored)" << endl;
  cerr << "   -kscw          constrained enumeration with k-sorted branching" << endl;
  cerr << "   -top <file>    specify a parameter file" << endl;
  cerr << "      --PARAMETER_NAME value   overrides a parameter" << endl << endl;
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
    bool ucwflag = args.getSwitch("-ucw", true);
    bool kscwflag = args.getSwitch("-kscw", true);
    bool crcwflag = args.getSwitch("-crcw", true);
    if (crcwflag) throw string("-crcw: this experimental enumerator is not available on this engine");

    Gn2Params ali_params;
    ApplicationParams app_params;
    RCfile default_rc;
    default_rc >> ali_params >> app_params;
    if (!topfile.empty()) {
      RCfile top_rc(topfile);
      top_rc >> ali_params >> app_params;
    }
    args >> ali_params >> app_params;
    if (args.count() != 2 && args.count() != 3) usage();

    cerr << "Reading in query profile...  ";
    HMAPSequence query(args.getArg(0).str().c_str());
    cerr << "length " << query.seq_length << endl;
    cerr << "Reading in template profile...  ";
    SMAPSequence templ(args.getArg(1).str().c_str());
    cerr << "length " << templ.seq_length << endl;

    Hmap2Eval ge(ali_params);
    DPMatrix<HMAPSequence, SMAPSequence, Hmap2Eval> dpm(query, templ, ge, fwd);
    clock_t t1 = clock();

    Optimal<HMAPSequence, SMAPSequence, Hmap2Eval> opt;
    AlignmentSet<HMAPSequence, SMAPSequence, Hmap2Eval> alignments(dpm, opt);
    cerr << "Added optimal alignment to alignment set." << endl;

    if (!optflag) {
      if (ucwflag) {
        cerr << "Now adding unconstrained suboptimal alignments." << endl;
        UnconstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> ucw(ali_params);
        ucw.enumerate(dpm, alignments);
      } else if (kscwflag) {
        cerr << "Now adding constrained suboptimal alignments, with branching limited by k-sort." << endl;
        SuboptFlags subopt(true, templ.size());
        templ.getDefaultFlags(subopt);
        if (args.count() > 2) {
          ifstream fin(args.getArg(2).str().c_str());
          fin >> Formats::FastaIn("Flags=suboptimal region", false) >> subopt;
        }
        KSConstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> kscno(ali_params, subopt);
        kscno.enumerate(dpm, alignments);
      } else {
        cerr << "Now adding constrained suboptimal alignments." << endl;
        SuboptFlags subopt(true, templ.size());
        templ.getDefaultFlags(subopt);
        if (args.count() > 2) {
          ifstream fin(args.getArg(2).str().c_str());
          fin >> Formats::FastaIn("Flags=suboptimal region", false) >> subopt;
        }
        ConstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> cno(ali_params, subopt);
        cno.enumerate(dpm, alignments);
      }
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
      case oHMAP:
        cout << Formats::HMAPOut(ali_params.submatrix_fn.c_str(), app_params.line_length) << alignments;
        break;
    }
    cerr << endl;
    cerr << "time for alignment was (sec) " << (t2 - t1) / (double)CLOCKS_PER_SEC << endl;
    cerr << "total cpu time was (sec) " << (t2 - t0) / (double)CLOCKS_PER_SEC << endl << endl;
  } catch (string e) {
    cerr << e << endl;
    exit(-1);
  }
  return 0;
}
