// hmap2_eval.h — see hmap_eval.h (Hmap2Eval lives there with HMAPaliEval; the reference keeps two identical copies)
#include "hmap_eval.h"
