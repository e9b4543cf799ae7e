// KPopCountDB -- drop-in for the reference's bin/KPopCountDB.ml: collects k-mer spectra into a
// database, combines them into class representatives, and exports tables / spectra / spectral distances.
//
// Same registers (database + selection), same actions executed in order of specification
// (bin/KPopCountDB.ml:94-330,357-436), same file naming and text formats (lib/KMerDB.ml).  Every loop over all
// counts runs in libkpop_hip.so: statistics, mean/median combination, transformations, distances.
//
// Differences from the reference, on purpose:
//   * -d/--distill is refused: it leans on BiOCamLib's OnlineStats/LinearFit, which are not part of the checkout;
//   * -T is accepted and ignored (the GPU is the parallelism);
//   * a runtime failure exits with status 1 (the reference prints the exception and exits 0);
//   * regular expressions are translated from OCaml Str syntax to ECMAScript (std::regex).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <functional>
#include <set>
#include <string>
#include <vector>

#include "../../include/kpop_hip.h"
#include "counter_db.h"

using namespace kpop_host;

namespace {

const char *kVersion = "49-hip";

void usage(FILE *f) {
  fprintf(f,
          "This is KPopCountDB (MI355X/HIP) version %s\n"
          "Usage: KPopCountDB [ACTIONS]   (executed in order of specification)\n"
          "Actions on the database register:\n"
          " -e|--empty                               put an empty database into the register\n"
          " -i|--input <prefix>                      load <prefix>.KPopCounter\n"
          " -m|--metadata|--add-metadata <file>      add metadata from a tab-separated table\n"
          " -k|--kmers|--add-kmers|--add-kmer-files <prefix>[,...]   add spectra from <prefix>.KPopSpectra.txt\n"
          " --combination-criterion|--spectrum-combination-criterion mean|median   (default mean)\n"
          " -c|--combine|--combine-by-class|--combine-spectra-by-class <metadata_field>\n"
          " --summary                                print a summary of the database\n"
          " -o|--output <prefix>                     save <prefix>.KPopCounter\n"
          " --distance|--distance-function euclidean|cosine|minkowski(<p>)   (default euclidean)\n"
          " --distance-normalize|--distance-normalization true|false   (default true)\n"
          " --distances|--compute-distances|--compute-spectral-distances SELECTOR SELECTOR <prefix>\n"
          "       SELECTOR := <metadata_field>~<regexp>[,...]; an empty field matches labels; writes <prefix>.KPopDMatrix\n"
          " --table-output-row-names|--table-output-col-names|--table-output-metadata|--table-transpose true|false\n"
          " --counts-threshold <x>  --counts-power <x>  --counts-transform|--counts-transformation binary|power|pseudocounts|clr\n"
          " --counts-output-zero-kmers|--counts-output-zero-k-mers true|false   --counts-precision <n>\n"
          " -t|--table|--to-table <prefix>           write <prefix>.KPopCounter.txt\n"
          " -s|--spectra|--to-spectra <prefix>       write <prefix>.KPopSpectra.txt\n"
          "Actions involving the selection register:\n"
          " -L|--labels|--selection-from-labels <label>[,...]\n"
          " -R|--regexps|--selection-from-regexps SELECTOR\n"
          " -A|--add-combined-selection|--selection-combine-and-add <label>\n"
          " -D|--delete|--selection-delete   -N|--selection-negate   -P|--selection-print   -C|--selection-clear\n"
          " -F|--selection-to-table-filter\n"
          "Miscellaneous: -T|--threads <n> (ignored)  -v|--verbose  -V|--version  -h|--help\n",
          kVersion);
}

[[noreturn]] void parse_error(const std::string &msg) {
  usage(stderr);
  fprintf(stderr, "(KPopCountDB): ERROR: %s\n", msg.c_str());
  exit(1);
}

bool parse_bool(const std::string &opt, const std::string &s) {
  if (s == "true") return true;
  if (s == "false") return false;
  parse_error("Option '" + opt + "': '" + s + "' is not a boolean");
}

double parse_float_non_neg(const std::string &opt, const std::string &s) {
  char *end = nullptr;
  const double v = strtod(s.c_str(), &end);
  if (end == s.c_str() || *end != 0 || !(v >= 0.)) parse_error("Option '" + opt + "': '" + s + "' is not a non-negative number");
  return v;
}

struct Distance {
  int kind = KPOP_EUCLIDEAN;
  double p = 2.;
};

Distance parse_distance(const std::string &s) {  // Space.Distance.of_string, lib/Space.ml:144-158
  Distance d;
  if (s == "euclidean") return d;
  if (s == "cosine") {
    d.kind = KPOP_COSINE;
    return d;
  }
  double p;
  char tail;
  if (sscanf(s.c_str(), "minkowski(%lf%c", &p, &tail) == 2 && tail == ')' && s.back() == ')' && p > 0.) {
    d.kind = KPOP_MINKOWSKI;
    d.p = p;
    return d;
  }
  throw Error("Unknown_distance(\"" + s + "\")");
}

std::vector<std::string> split_commas(const std::string &s) {
  std::vector<std::string> out;
  size_t at = 0;
  for (;;) {
    const size_t c = s.find(',', at);
    out.push_back(s.substr(at, c == std::string::npos ? std::string::npos : c - at));
    if (c == std::string::npos) break;
    at = c + 1;
  }
  return out;
}

struct State {
  CounterDB db;
  std::set<std::string> selected;
  int criterion = KPOP_COMBINE_MEAN;
  TableFilter filter;
  Distance distance;
  bool distance_normalise = true;
};

}  // namespace

int main(int argc, char **argv) {
  std::vector<std::function<void(State &)>> program;
  bool verbose = false;
  auto need = [&](int &i, const std::string &opt) -> std::string {
    if (i + 1 >= argc) parse_error("Option '" + opt + "' needs a parameter");
    return argv[++i];
  };
  auto selector = [&](const std::string &opt, const std::string &s) {
    try {
      return parse_regexp_selector(s);
    } catch (const std::exception &e) {
      parse_error("Option '" + opt + "': " + e.what());
    }
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto is = [&](std::initializer_list<const char *> names) {
      for (const char *n : names)
        if (a == n) return true;
      return false;
    };
    if (is({"-e", "--empty"})) program.push_back([](State &s) { s.db = CounterDB(); });
    else if (is({"-i", "--input"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db = CounterDB::of_binary(p); });
    } else if (is({"-m", "--metadata", "--add-metadata"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db.add_meta(p); });
    } else if (is({"-k", "--kmers", "--add-kmers", "--add-kmer-files"})) {
      const std::vector<std::string> p = split_commas(need(i, a));
      program.push_back([p](State &s) { s.db.add_files(p); });
    } else if (is({"--combination-criterion", "--spectrum-combination-criterion"})) {
      const std::string c = need(i, a);
      int crit;
      if (c == "mean") crit = KPOP_COMBINE_MEAN;
      else if (c == "median") crit = KPOP_COMBINE_MEDIAN;
      else parse_error("Unknown_combination_criterion(\"" + c + "\")");
      program.push_back([crit](State &s) { s.criterion = crit; });
    } else if (is({"-c", "--combine", "--combine-by-class", "--combine-spectra-by-class"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db.split_spectra(p, s.criterion); });
    } else if (is({"-d", "--distill", "--distill-kmers"})) {
      need(i, a);
      need(i, a);
      program.push_back([](State &) { throw Error("-d/--distill is not supported by the HIP build (see the head of KPopCountDB.cpp)"); });
    } else if (is({"--summary"})) program.push_back([](State &s) { s.db.output_summary(); });
    else if (is({"-o", "--output"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db.to_binary(p); });
    } else if (is({"--distance", "--distance-function"})) {
      Distance d;
      try {
        d = parse_distance(need(i, a));
      } catch (const std::exception &e) {
        parse_error(e.what());
      }
      program.push_back([d](State &s) { s.distance = d; });
    } else if (is({"--distance-normalize", "--distance-normalization"})) {
      const bool b = parse_bool(a, need(i, a));
      program.push_back([b](State &s) { s.distance_normalise = b; });
    } else if (is({"--distances", "--compute-distances", "--compute-spectral-distances"})) {
      const RegexpSelector r1 = selector(a, need(i, a)), r2 = selector(a, need(i, a));
      const std::string p = need(i, a);
      program.push_back([r1, r2, p](State &s) {
        s.db.to_distances(s.distance.kind, s.distance.p, s.distance_normalise, s.db.selected_from_regexps(r1), s.db.selected_from_regexps(r2), p);
      });
    } else if (is({"--table-output-row-names"})) {
      const bool b = parse_bool(a, need(i, a));
      program.push_back([b](State &s) { s.filter.print_row_names = b; });
    } else if (is({"--table-output-col-names"})) {
      const bool b = parse_bool(a, need(i, a));
      program.push_back([b](State &s) { s.filter.print_col_names = b; });
    } else if (is({"--table-output-metadata"})) {
      const bool b = parse_bool(a, need(i, a));
      program.push_back([b](State &s) { s.filter.print_metadata = b; });
    } else if (is({"--table-transpose"})) {
      const bool b = parse_bool(a, need(i, a));
      program.push_back([b](State &s) { s.filter.transpose = b; });
    } else if (is({"--counts-threshold"})) {
      const double v = parse_float_non_neg(a, need(i, a));
      program.push_back([v](State &s) { s.filter.transform.threshold = v; });
    } else if (is({"--counts-power"})) {
      const double v = parse_float_non_neg(a, need(i, a));
      program.push_back([v](State &s) { s.filter.transform.power = v; });
    } else if (is({"--counts-transform", "--counts-transformation"})) {
      const std::string w = need(i, a);
      program.push_back([w](State &s) {
        s.filter.transform.which = w;
        (void)s.filter.transform.code();  // Unknown_transformation is raised when the action runs (bin/KPopCountDB.ml:404-406)
      });
    } else if (is({"--counts-output-zero-kmers", "--counts-output-zero-k-mers"})) {
      const bool b = parse_bool(a, need(i, a));
      program.push_back([b](State &s) { s.filter.print_zero_rows = b; });
    } else if (is({"--counts-precision"})) {
      const int v = atoi(need(i, a).c_str());
      if (v <= 0) parse_error("Option '" + a + "': precision must be a positive integer");
      program.push_back([v](State &s) { s.filter.precision = v; });
    } else if (is({"-t", "--table", "--to-table"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db.to_table(s.filter, p); });
    } else if (is({"-s", "--spectra", "--to-spectra"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db.to_spectra(s.filter, p); });
    } else if (is({"-L", "--labels", "--selection-from-labels"})) {
      const std::string labels = need(i, a);
      if (!labels.empty()) {  // an empty list is no action at all (bin/KPopCountDB.ml:275-278)
        const std::vector<std::string> l = split_commas(labels);
        program.push_back([l](State &s) { s.selected = std::set<std::string>(l.begin(), l.end()); });
      }
    } else if (is({"-R", "--regexps", "--selection-from-regexps"})) {
      const RegexpSelector r = selector(a, need(i, a));
      program.push_back([r](State &s) { s.selected = s.db.selected_from_regexps(r); });
    } else if (is({"-A", "--add-combined-selection", "--selection-combine-and-add"})) {
      const std::string p = need(i, a);
      program.push_back([p](State &s) { s.db.add_combined_selected(p, s.selected, s.criterion); });
    } else if (is({"-D", "--delete", "--selection-delete"})) program.push_back([](State &s) { s.db.remove_selected(s.selected); });
    else if (is({"-N", "--selection-negate"})) program.push_back([](State &s) { s.selected = s.db.selected_negate(s.selected); });
    else if (is({"-P", "--selection-print"}))
      program.push_back([](State &s) {
        fprintf(stderr, "Currently selected spectra = [");
        for (const std::string &l : s.selected) fprintf(stderr, " '%s'", l.c_str());
        fprintf(stderr, " ].\n");
      });
    else if (is({"-C", "--selection-clear"})) program.push_back([](State &s) { s.selected.clear(); });
    else if (is({"-F", "--selection-to-table-filter"})) program.push_back([](State &s) { s.filter.filter_columns = s.selected; });
    else if (is({"-T", "--threads"})) {
      if (atoi(need(i, a).c_str()) <= 0) parse_error("Option '" + a + "': the number of threads must be positive");
    } else if (is({"-v", "--verbose"})) verbose = true;
    else if (is({"-V", "--version"})) {
      printf("%s\n", kVersion);
      return 0;
    } else if (is({"-x", "--print-exception-backtrace"})) {
    } else if (is({"-h", "--help"})) {
      usage(stdout);
      return 0;
    } else {
      parse_error("Unknown option '" + a + "'");
    }
  }
  if (program.empty()) {  // bin/KPopCountDB.ml:332-335
    usage(stdout);
    return 0;
  }
  try {
    State st;
    stage_mark("KPopCountDB", "start");
    for (auto &action : program) {
      st.db.verbose = verbose;
      action(st);
      stage_mark("KPopCountDB", "action done");
    }
    fflush(stdout);
    fflush(stderr);
    _exit(0);  // every output is written and closed: the database need not be taken apart first
  } catch (const std::exception &e) {
    fprintf(stderr, "(KPopCountDB): FATAL: Uncaught exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
