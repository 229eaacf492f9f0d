#!/usr/bin/env python3
"""Writes the SURVEY.md Appendix D known-answer vector as fixture files.

Provenance: the 15 rows, 3 reads and expected .pml/.cid text below are
transcribed from SURVEY.md Appendix D, which records output of the reference
`pml_query` captured during the survey.  It is the only reference-produced
vector available (the reference ships no fixtures).
Files written next to this script: kat_d.col_pml, kat_d.fa, kat_d.fa.pml,
kat_d.fa.cid.
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

ROWS_HEX = """
41 00 00 00 00 00 01 00 00 00 00 00 00 00 00 00 00 00
43 01 00 00 00 00 09 00 00 00 00 00 02 00 00 00 00 00
54 02 00 00 00 00 0c 00 00 00 01 00 04 00 00 00 00 00
41 03 00 00 00 00 02 00 00 00 00 00 01 01 00 00 00 00
54 04 00 00 00 00 0d 00 00 00 00 00 03 03 00 00 00 00
54 05 00 00 00 00 0d 00 00 00 01 00 00 03 00 00 00 00
43 06 00 00 00 00 09 00 00 00 01 00 02 02 00 00 00 00
47 07 00 00 00 00 0a 00 00 00 01 00 04 00 00 00 00 00
47 08 00 00 00 00 0b 00 00 00 00 00 01 00 00 00 00 00
41 0a 00 00 00 00 03 00 00 00 00 00 03 0a 00 00 00 00
43 0d 00 00 00 00 09 00 00 00 02 00 00 0a 00 00 00 00
01 0f 00 00 00 00 00 00 00 00 00 00 02 00 00 00 00 00
41 10 00 00 00 00 06 00 00 00 00 00 04 0e 00 00 00 00
54 12 00 00 00 00 0e 00 00 00 00 00 01 0a 00 00 00 00
41 14 00 00 00 00 08 00 00 00 00 00 03 14 00 00 00 00
"""

FASTA = b">q1 desc\nGATTACA\n>q2\nTTACCGATNACA\n>q3\nCCCC\n"
PML = b">q1 \n5 4 3 2 1 0 1 \n>q2 \n4 3 2 1 0 3 2 1 0 1 0 1 \n>q3 \n0 0 1 0 \n"
CID = b">q1 \n1 3 1 3 3 1 3 \n>q2 \n1 0 3 0 2 1 3 1 3 3 1 3 \n>q3 \n3 3 0 3 \n"


def main():
    rows = bytes(int(t, 16) for t in ROWS_HEX.split())
    assert len(rows) == 15 * 18
    blob = struct.pack("<4Q", 13, 22, 15, 15) + rows   # bwt_r, n, r, size
    assert len(blob) == 302
    for name, data in (("kat_d.col_pml", blob), ("kat_d.fa", FASTA),
                       ("kat_d.fa.pml", PML), ("kat_d.fa.cid", CID)):
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(data)


if __name__ == "__main__":
    main()
