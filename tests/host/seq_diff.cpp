// Test harness (not product): reads argv[1] (format argv[2] = fasta|fastq) with the line-by-line SeqReader and with
// the block-wise FastSeqReader; they must agree on success or failure and on every tag and linted sequence.  The
// FastSeqReader's batches also go through the reads-stream writer and reader (a pipe) and must come back unchanged.
// Exit 0 = agree, 2 = only one failed / messages differ, 3 = content differs.
#include <stdio.h>
#include <unistd.h>

#include <string>
#include <thread>

#include "../../kpop_amd/host/fast_seq.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  if (argc < 3) return 64;
  const SeqFormat fmt = std::string(argv[2]) == "fastq" ? SeqFormat::FASTQ : SeqFormat::FASTA;
  std::string e1, e2;
  std::vector<std::string> tags1, seqs1, tags2, seqs2, tags3, seqs3;
  try {
    SeqReader r(argv[1], fmt);
    std::string t, s;
    while (r.next_record(t, s)) {
      tags1.push_back(t);
      seqs1.push_back(s);
    }
  } catch (const std::exception &e) {
    e1 = e.what();
  }
  int fds[2];
  if (pipe(fds) != 0) return 65;
  std::thread consumer([&] {
    try {
      char magic[8];
      if (!read_all(fds[0], magic, 8)) return;
      ReadStreamReader rs(fds[0]);
      FlatBatch b;
      while (rs.next(b)) {
        size_t ob = 0, ot = 0;
        for (size_t i = 0; i < b.size(); ++i) {
          tags3.emplace_back(b.tags.data() + ot, b.tag_lens[i]);
          seqs3.emplace_back((const char *)b.bases.data() + ob, b.lens[i]);
          ob += b.lens[i];
          ot += b.tag_lens[i];
        }
      }
    } catch (const std::exception &e) {
      tags3.assign(1, std::string("STREAM ERROR ") + e.what());
    }
  });
  try {
    FastSeqReader r(argv[1], fmt);
    FlatBatch b;
    ReadStreamHeader h;
    h.k = 12;
    write_stream_header(fds[1], h);
    while (r.next(b)) {
      if (b.size() == 0) {
        e2 = "EMPTY BATCH";
        break;
      }
      write_stream_block(fds[1], b);
      size_t ob = 0, ot = 0;
      for (size_t i = 0; i < b.size(); ++i) {
        tags2.emplace_back(b.tags.data() + ot, b.tag_lens[i]);
        seqs2.emplace_back((const char *)b.bases.data() + ob, b.lens[i]);
        ob += b.lens[i];
        ot += b.tag_lens[i];
      }
    }
  } catch (const std::exception &e) {
    e2 = e.what();
  }
  write_stream_end(fds[1]);
  close(fds[1]);
  consumer.join();
  if (e1 != e2) {
    printf("MESSAGES DIFFER: [%s] vs [%s]\n", e1.c_str(), e2.c_str());
    return 2;
  }
  if (!e1.empty()) return 0;
  if (tags1 != tags2 || seqs1 != seqs2) {
    printf("CONTENT DIFFERS: %zu vs %zu records\n", tags1.size(), tags2.size());
    return 3;
  }
  if (tags3 != tags2 || seqs3 != seqs2) {
    printf("STREAM ROUND TRIP DIFFERS: %zu vs %zu records (%s)\n", tags3.size(), tags2.size(), tags3.empty() ? "" : tags3[0].c_str());
    return 3;
  }
  return 0;
}
