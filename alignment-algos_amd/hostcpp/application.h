// application.h — output parameters (reference application.h:17-44, application.cpp:16-48).
#ifndef ALN_HOST_APPLICATION_H
#define ALN_HOST_APPLICATION_H
#include <string>
#include "pstore.h"

enum output_format_t { oHMAP = 0, oPIR = 1, oFASTA = 2 };

class ApplicationParams {
 public:
  ApplicationParams() : output_format(oFASTA), line_length(60), verbosity(0), log_file("") {}
  void read(ParamStore* p) {
    std::string s;
    s = "OUTPUT_FORMAT";
    if (p->find(s)) { int v = output_format; p->getValue(s) >> v; output_format = static_cast<output_format_t>(v); }
    s = "OUTPUT_LINE_LENGTH";
    if (p->find(s)) p->getValue(s) >> line_length;
    s = "VERBOSE";
    if (p->find(s)) p->getValue(s) >> verbosity;
    s = "LOG_FILE";
    if (p->find(s)) p->getValue(s) >> log_file;
  }
  output_format_t output_format;
  int line_length;
  int verbosity;
  std::string log_file;
};
#endif
