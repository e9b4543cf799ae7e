// KPopTwist_ -- drop-in for the reference's bin/KPopTwist_.ml: parses KPopTwist's command line and echoes it as one
// line of \\001-separated fields (:136-140), which is all the reference's bash wrapper src/KPopTwist wants from it
// (:19-27).  With this, KPopCountDB, KPopTwistDB and KPopTwistCA (in place of `Rscript`, one word changed at :49) on
// the PATH, that wrapper runs as it stands; kpop_amd/bin/KPopTwist does the same work in one process.
#include <stdio.h>

#include "twist_args.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  const TwistArgs A = parse_twist_args(argc, argv, "KPopTwist", "27-hip");
  // "%s\001%s\001%.12g\001%.12g\001%.12g\001%s\001%b\001%.12g\001%s\001%s\001%d\001%b\001%b\n"
  printf("%s\001%s\001%.12g\001%.12g\001%.12g\001%s\001%s\001%.12g\001%s\001%s\001%d\001%s\001%s\n", A.input.c_str(),
         A.ca.keep_path.c_str(), A.ca.fraction, A.transform.threshold, A.transform.power, A.transform.which.c_str(),
         A.ca.normalize ? "true" : "false", A.ca.threshold, A.output.c_str(), A.output_kmers.c_str(), A.threads,
         A.temporaries ? "true" : "false", A.ca.verbose ? "true" : "false");
  return 0;
}
