#!/usr/bin/env python3
"""File-to-file rate of the drop-in binaries: FASTA on disk -> .KPopTwisted / .KPopSummary.txt, the README's commands
(README.md:606,656) typed as the README types them.

    python3 tools/file_to_file.py --reads 1000000 -k 12 [--json] [--keep DIR]

Untimed set-up: C class genomes (synthetic, 30 kb) are counted, combined and turned into a twister by the training
commands of README.md:91-93 (KPopCount | KPopCountDB, KPopTwist), and the reads are sampled from those genomes with
0.5 % substitutions (tools synth: kpop_amd/bin/kpop_synth), so that -- as with real data -- most of a read's k-mers are
columns of the twister.  Timed, each as wall time of the whole shell pipeline, page cache warm:

  A  KPopCount -k K -L -f reads.fa | KPopTwistDB -i T Classes -k /dev/stdin -o t Test          (README.md:606)
  A' the same with KPOP_PIPE_FORMAT=text: the spectra cross the pipe as text, the way the reference does it
  B  KPopTwistDB -i T Classes -i t Classes -s Test Summary                                      (README.md:656)

A and A' must produce byte-identical Test.KPopTwisted files; the JSON line (bench.py's file_to_file object) reports A + B.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kpop_amd", "bin")


def sh(cmd, env=None, cwd=None):
    t0 = time.perf_counter()
    r = subprocess.run(["bash", "-c", "set -o pipefail; " + cmd], cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("command failed (%d): %s\n%s" % (r.returncode, cmd, r.stderr.decode("utf-8", "replace")[-2000:]))
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("--classes", type=int, default=65)
    ap.add_argument("--class-len", type=int, default=30000)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--keep", default=None, help="work in this directory and leave the files there")
    ap.add_argument("--skip-text", action="store_true", help="do not time the text-spectra variant A'")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--input-format", choices=("fasta", "fastq", "paired"), default="fasta",
                    help="the reads as FASTA (-f), as FASTQ (-s) or as two FASTQ files of mates (-p)")
    args = ap.parse_args()

    env = dict(os.environ)
    env["PATH"] = BIN + os.pathsep + env.get("PATH", "")
    base = args.keep or tempfile.mkdtemp(prefix="kpop_f2f_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.makedirs(base, exist_ok=True)
    log = (lambda *a: None) if args.json else (lambda *a: print(*a, flush=True))
    try:
        K, C = args.k, args.classes
        t = sh("kpop_synth genomes --n %d --len %d --seed 12648430 > classes.fa" % (C, args.class_len), env, base)
        t += sh("kpop_synth reads --from classes.fa --n %d --len %d --mutate 0.005 --seed 1263555440 > reads.fa" % (args.reads, args.read_len), env, base)
        log("set-up: %d class genomes of %d bp, %d reads of %d bp sampled from them: %.2f s" % (C, args.class_len, args.reads, args.read_len, t))
        # training, README.md:91-93: one spectrum per class genome (-L on the multi-FASTA does what 65 runs of -l do)
        t = sh("KPopCount -k %d -L -f classes.fa | KPopCountDB -k /dev/stdin -o Classes" % K, env, base)
        t2 = sh("KPopTwist -i Classes -o Classes", env, base)
        sizes = {f: os.path.getsize(os.path.join(base, f)) for f in ("reads.fa", "Classes.KPopTwister", "Classes.KPopTwisted")}
        log("training (untimed part): count+combine %.2f s, KPopTwist %.2f s; twister file %.1f MB; reads.fa %.1f MB"
            % (t, t2, sizes["Classes.KPopTwister"] / 1e6, sizes["reads.fa"] / 1e6))
        reads_arg = "-f reads.fa"
        if args.input_format != "fasta":  # the same reads as FASTQ, or dealt to two files of mates
            with open(os.path.join(base, "reads.fa")) as f, open(os.path.join(base, "reads.fq"), "w") as one, \
                    open(os.path.join(base, "mates_1.fq"), "w") as m1, open(os.path.join(base, "mates_2.fq"), "w") as m2:
                i = 0
                while True:
                    h = f.readline()
                    if not h:
                        break
                    sq = f.readline().rstrip("\n")
                    rec = "@" + h[1:] + sq + "\n+\n" + "I" * len(sq) + "\n"
                    if args.input_format == "fastq":
                        one.write(rec)
                    else:
                        (m1 if i % 2 == 0 else m2).write(rec)
                    i += 1
            reads_arg = "-s reads.fq" if args.input_format == "fastq" else "-p mates_1.fq mates_2.fq"
        cmd_a = "KPopCount -k %d -L %s | KPopTwistDB -i T Classes -k /dev/stdin -o t Test" % (K, reads_arg)
        cmd_b = "KPopTwistDB -i T Classes -i t Classes -s Test Summary"
        sh(cmd_a, env, base)  # page cache and code objects warm
        ta = min(sh(cmd_a, env, base) for _ in range(args.reps))
        tb = min(sh(cmd_b, env, base) for _ in range(args.reps))
        out_bytes = os.path.getsize(os.path.join(base, "Test.KPopTwisted"))
        res = {"value": args.reads / (ta + tb), "unit": "sequences/sec", "reads": args.reads, "k": K,
               "seconds": {"count|twist -> Test.KPopTwisted": ta, "distance summary -> Summary.KPopSummary.txt": tb},
               "count_twist_only": args.reads / ta,
               "bytes": {"reads.fa": sizes["reads.fa"], "Classes.KPopTwister": sizes["Classes.KPopTwister"], "Test.KPopTwisted": out_bytes},
               "commands": [cmd_a, cmd_b],
               "note": "wall time of the README's shell pipelines through the drop-in binaries, page cache warm, best of %d; "
                       "each command pays its own process start, HIP bring-up and twister load" % args.reps}
        if not args.skip_text:
            shutil.copy(os.path.join(base, "Test.KPopTwisted"), os.path.join(base, "Test_reads.KPopTwisted"))
            envt = dict(env, KPOP_PIPE_FORMAT="text")
            tt = sh(cmd_a, envt, base)
            same = open(os.path.join(base, "Test.KPopTwisted"), "rb").read() == open(os.path.join(base, "Test_reads.KPopTwisted"), "rb").read()
            res["text_spectra_variant"] = {"seconds": tt, "sequences_per_sec": args.reads / tt, "byte_identical_output": same}
            if not same:
                raise RuntimeError("the reads-stream pipeline and the text-spectra pipeline wrote different Test.KPopTwisted files")
        if args.json:
            print(json.dumps(res))
        else:
            print(json.dumps(res, indent=1))
    finally:
        if not args.keep:
            shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
