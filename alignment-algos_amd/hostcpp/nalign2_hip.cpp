// nalign2_hip.cpp — profile-profile alignment driver on the MI355X engine (the job the reference's nalign2 does:
// nalign2.cpp, make target `nalign2`): a query HMAP profile against a template profile with Hmap2Eval, a global DPMatrix,
// the Optimal alignment, then one near-optimal enumerator chosen on the command line, written as FASTA / PIR / HMAP.
//
// Own structure: the command line becomes a Job, the enumerator is picked from a table (one row per switch), and the three
// stages (load, align, write) are separate functions.  The parameter layering is the reference's: programmed defaults,
// ~/.hmaprc, the -top file, then --KEY value overrides.  What differs from the reference binary: the template is read as an
// HMAP profile (the reference's SMAPSequence also loads a PDB structure through the Troll library, which Hmap2Eval never looks
// at); -kscw / -crcw run the engine's KSConstrainedNearOptimal / CRConstrainedNearOptimal, whose parity is unpinned (kscw.h
// and crcw.h do not compile on LP64).  The reference binary cannot be built here, so this driver's stdout
// has no golden; its pieces (HMAP parser, Hmap2Eval, DPMatrix, Optimal, cw, writers) are each checked on their own.
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>

#include "application.h"
#include "argv.h"
#include "crcw.h"
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

namespace {

typedef DPMatrix<HMAPSequence, SMAPSequence, Hmap2Eval> ProfileMatrix;
typedef AlignmentSet<HMAPSequence, SMAPSequence, Hmap2Eval> ProfileAlignments;

enum Search { kConstrained, kOptimalOnly, kUnconstrained, kKSorted, kClusterReduced };

// one row per enumerator switch; the first switch present on the command line wins, none = the constrained search
const struct { const char* flag; Search search; bool uses_flags; const char* what; } kSearches[] = {
    {"-opt", kOptimalOnly, false, "optimal alignment only"},
    {"-ucw", kUnconstrained, false, "unconstrained near-optimal alignments"},
    {"-kscw", kKSorted, true, "constrained near-optimal alignments, k best branches per node"},
    {"-crcw", kClusterReduced, true, "constrained near-optimal alignments, overlap-reduced in rounds"},
};

struct Job {
  Gn2Params scoring;
  ApplicationParams output;
  Search search;
  bool uses_flags;
  std::string what;
  std::string query_file, template_file, flag_file;
};

void print_usage_and_exit() {
  std::cerr << "\nUsage: nalign2_hip [switch] [-top file] [--KEY value ...] query.prof template.prof [template.flag]\n\n"
            << "   Optimal and near-optimal profile-profile alignments (Hmap2Eval), computed on an MI355X.\n"
            << "   template.flag  regions of the template in which alignments may branch (default: everything but loops)\n";
  for (const auto& s : kSearches) std::cerr << "   " << s.flag << "\t" << s.what << "\n";
  std::cerr << "   -top <file>    parameter file (KEY: value lines); --KEY value overrides a single parameter\n\n";
  exit(0);
}

Job read_command_line(int argc, const char** argv) {
  if (argc == 0) print_usage_and_exit();
  Argv args(argc, argv);
  if (args.help()) print_usage_and_exit();
  Job job;
  job.search = kConstrained; job.uses_flags = true; job.what = "constrained near-optimal alignments";
  std::string top;
  if (args.getSwitch("-top", false)) args.getSwitch("-top", 1) >> top;
  bool chosen = false;
  for (const auto& s : kSearches)
    if (args.getSwitch(s.flag, true) && !chosen) { job.search = s.search; job.uses_flags = s.uses_flags; job.what = s.what; chosen = true; }
  RCfile home;                                   // ~/.hmaprc; a missing file only warns
  home >> job.scoring >> job.output;
  if (!top.empty()) { RCfile extra(top); extra >> job.scoring >> job.output; }
  args >> job.scoring >> job.output;
  if (args.count() < 2 || args.count() > 3) print_usage_and_exit();
  job.query_file = args.getArg(0).str();
  job.template_file = args.getArg(1).str();
  if (args.count() == 3) job.flag_file = args.getArg(2).str();
  return job;
}

// where alignments may branch: the template's own default (no loops), replaced by a flag file when one is given
SuboptFlags branch_regions(SMAPSequence& templ, const std::string& flag_file) {
  SuboptFlags regions(true, templ.size());
  templ.getDefaultFlags(regions);
  if (!flag_file.empty()) {
    std::ifstream in(flag_file.c_str());
    in >> Formats::FastaIn("Flags=suboptimal region", false) >> regions;
  }
  return regions;
}

void add_near_optimal(Job& job, SMAPSequence& templ, ProfileMatrix& dpm, ProfileAlignments& found) {
  if (job.search == kOptimalOnly) return;
  std::cerr << "Adding " << job.what << "." << std::endl;
  if (job.search == kUnconstrained) {
    UnconstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> e(job.scoring);
    e.enumerate(dpm, found);
    return;
  }
  SuboptFlags regions = branch_regions(templ, job.flag_file);
  switch (job.search) {
    case kKSorted: { KSConstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> e(job.scoring, regions); e.enumerate(dpm, found); break; }
    case kClusterReduced: { CRConstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> e(job.scoring, regions); e.enumerate(dpm, found); break; }
    default: { ConstrainedNearOptimal<HMAPSequence, SMAPSequence, Hmap2Eval> e(job.scoring, regions); e.enumerate(dpm, found); break; }
  }
}

void write_alignments(const Job& job, ProfileAlignments& found) {
  const int width = job.output.line_length;
  if (job.output.output_format == oPIR) std::cout << Formats::PIROut(width) << found;
  else if (job.output.output_format == oHMAP) std::cout << Formats::HMAPOut(job.scoring.submatrix_fn.c_str(), width) << found;
  else std::cout << Formats::FastaOut(width) << found;
}

double seconds(clock_t a, clock_t b) { return (b - a) / (double)CLOCKS_PER_SEC; }

}  // namespace

int main(int argc, const char** argv) {
  try {
    const clock_t started = clock();
    Job job = read_command_line(argc, argv);
    HMAPSequence query(job.query_file.c_str());
    SMAPSequence templ(job.template_file.c_str());
    std::cerr << "query profile: " << query.seq_length << " residues; template profile: " << templ.seq_length << " residues" << std::endl;

    Hmap2Eval scorer(job.scoring);
    ProfileMatrix dpm(query, templ, scorer, fwd);            // global by default, like the reference's constructor call
    const clock_t built = clock();
    Optimal<HMAPSequence, SMAPSequence, Hmap2Eval> best;
    ProfileAlignments found(dpm, best);
    add_near_optimal(job, templ, dpm, found);
    found.assignIdentity();
    const clock_t aligned = clock();

    write_alignments(job, found);
    std::cerr << "\nalignment (traceback + enumeration): " << seconds(built, aligned) << " s; whole run: " << seconds(started, aligned)
              << " s of cpu time\n" << std::endl;
  } catch (std::string e) {
    std::cerr << e << std::endl;
    return 255;
  }
  return 0;
}
