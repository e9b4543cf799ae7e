// twist_args.h -- the command line of KPopTwist (bin/KPopTwist_.ml:52-135), shared by KPopTwist (which acts on it)
// and KPopTwist_ (which, like the reference's, only echoes it for the bash wrapper src/KPopTwist).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <thread>

#include "ca_pipeline.h"
#include "counter_db.h"

namespace kpop_host {

struct TwistArgs {
  std::string input, output, output_kmers;
  Transform transform;
  CaParams ca;
  int threads = (int)std::max(1u, std::thread::hardware_concurrency());  // Processes.Parallel.get_nproc ()
  bool temporaries = false;
};

inline void twist_usage(FILE *f, const char *name, const char *version) {
  fprintf(f,
          "This is %s (MI355X/HIP) version %s\n"
          "Usage: %s -i|--input <binary_input_prefix> -o|--output <binary_output_prefix> [OPTIONS]\n"
          " -k|--kmers|--keep|--keep-kmers|--kmers-keep <file>   keep only the k-mers listed (one per line)\n"
          " -s|--sample|--sample-kmers|--kmers-sample <fraction>   resample k-mers (default 1)\n"
          " --counts-threshold <x>   --counts-power <x>   --counts-transform|--counts-transformation binary|power|pseudocounts|clr\n"
          " --counts-normalize|--counts-normalization true|false   (default true)\n"
          " --kmers-threshold <x>    drop k-mers whose total is below x times the largest total (default 0)\n"
          " -i|--input <prefix>      <prefix>.KPopCounter\n"
          " -o|--output <prefix>     <prefix>.KPopTwister and <prefix>.KPopTwisted\n"
          " -K|--output-kmers|--output-twisted-kmers <prefix>   also save the twisted k-mers\n"
          " -T|--threads <n>  --keep-temporaries  -v|--verbose  -V|--version  -h|--help\n",
          name, version, name);
}

// exits on -h / -V / a parse error, as Tools.Argv does
inline TwistArgs parse_twist_args(int argc, char **argv, const char *name, const char *version) {
  TwistArgs A;
  auto parse_error = [&](const std::string &msg) {
    twist_usage(stderr, name, version);
    fprintf(stderr, "(%s): ERROR: %s\n", name, msg.c_str());
    exit(1);
  };
  auto need = [&](int &i, const std::string &opt) -> std::string {
    if (i + 1 >= argc) parse_error("Option '" + opt + "' needs a parameter");
    return argv[++i];
  };
  auto number = [&](const std::string &opt, const std::string &s, double lo, double hi) {
    char *end = nullptr;
    const double v = strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != 0 || !(v >= lo) || !(v <= hi)) parse_error("Option '" + opt + "': '" + s + "' is out of range");
    return v;
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto is = [&](std::initializer_list<const char *> names) {
      for (const char *n : names)
        if (a == n) return true;
      return false;
    };
    if (is({"-k", "--kmers", "--keep", "--keep-kmers", "--kmers-keep"})) A.ca.keep_path = need(i, a);
    else if (is({"-s", "--sample", "--sample-kmers", "--kmers-sample"})) A.ca.fraction = number(a, need(i, a), 0., 1.);
    else if (is({"--counts-threshold"})) A.transform.threshold = number(a, need(i, a), 0., 1e300);
    else if (is({"--counts-power"})) A.transform.power = number(a, need(i, a), 0., 1e300);
    else if (is({"--counts-transform", "--counts-transformation"})) A.transform.which = need(i, a);
    else if (is({"--counts-normalize", "--counts-normalization"})) {
      const std::string b = need(i, a);
      if (b != "true" && b != "false") parse_error("Option '" + a + "': '" + b + "' is not a boolean");
      A.ca.normalize = b == "true";
    } else if (is({"--kmers-threshold"})) A.ca.threshold = number(a, need(i, a), 0., 1e300);
    else if (is({"-i", "--input"})) A.input = need(i, a);
    else if (is({"-o", "--output"})) A.output = need(i, a);
    else if (is({"-K", "--output-kmers", "--output-twisted-kmers"})) A.output_kmers = need(i, a);
    else if (is({"-T", "--threads"})) {
      A.threads = atoi(need(i, a).c_str());
      if (A.threads <= 0) parse_error("Option '" + a + "': the number of threads must be positive");
    } else if (is({"--keep-temporaries"})) A.temporaries = true;
    else if (is({"-v", "--verbose"})) A.ca.verbose = true;
    else if (is({"-V", "--version"})) {
      printf("%s\n", version);
      exit(0);
    } else if (is({"-h", "--help"})) {
      twist_usage(stdout, name, version);
      exit(0);
    } else {
      parse_error("Unknown option '" + a + "'");
    }
  }
  if (A.input.empty()) parse_error("Option '-i' is mandatory");   // TA.Mandatory, bin/KPopTwist_.ml:97-103
  if (A.output.empty()) parse_error("Option '-o' is mandatory");  // :104-110
  return A;
}

}  // namespace kpop_host
