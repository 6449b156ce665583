// hmap2_eval.h — see hmap_eval.h (Hmap2Eval lives there with HMAPaliEval; the reference keeps two identical copies).
// The reference's Hmap2Eval is constructed from a Gn2Params (hmap2_eval.h:25), so that class comes along.
#include "hmap_eval.h"
#include "gn2_eval.h"
