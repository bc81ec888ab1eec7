"""Register / scratch table of every kernel in a `-save-temps` gfx950 .s file (the amdhsa metadata block).
usage: python3 tools/isa_stats.py file.s [name-filter]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    meta = txt[txt.index("amdhsa.kernels:"):]
    rows = []
    for blk in meta.split("  - .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
        name = get("name")
        m = re.match(r"_ZN2wv\d+([a-z0-9_]+?)(I[A-Za-z0-9_]*?)?EE?vP", name)   # wv::name<template args>(...)
        if m:
            args = re.findall(r"Lb([01])E|Li(\d+)E|^I([fd])|(?<=I)([fd])", m.group(2) or "")
            name = m.group(1) + ("<" + ",".join(next(x for x in a if x) for a in args) + ">" if args else "")
        if flt in name:
            rows.append((name, get("vgpr_count"), get("agpr_count"), get("sgpr_count"), get("private_segment_fixed_size"),
                         get("vgpr_spill_count"), get("group_segment_fixed_size"), get("max_flat_workgroup_size")))
    print("%-70s %5s %5s %5s %8s %6s %7s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "spill", "lds", "wg"))
    for r in rows:
        print("%-70s %5s %5s %5s %8s %6s %7s %5s" % r)


main()
