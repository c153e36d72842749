#!/usr/bin/env python3
"""How integration/7bgzf-hip.patch's hunks for the six applets 7dictzip / 7razf / 7gzinga / 7png / 7ciso / 7daxcr were made
(round 5; the hunks for 7bgzf, 7migz, lib/zlibutil.[ch] and bgzf_compress.c were written by hand in round 4 and are kept).

Every applet of the reference has the same three dispatch ladders (method -> func, error names, popt table +
option loop + exactly-one check + final dispatch); this script makes the same edits in a scratch copy of each file under
oracle/_ref/hip_work/ (git-ignored build scratch -- the reference's sources are never copied into the repository's history)
and prints `diff -U1` of the result: the text appended to integration/7bgzf-hip.patch.  Usage, where /root/reference exists:

    python3 integration/make_patch_hunks.py >> integration/7bgzf-hip.patch     (after cutting the old six-applet part)
"""
import os
import re
import subprocess
import sys

REF = os.environ.get("REF", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = os.path.join(ROOT, "oracle", "_ref", "hip_work")
APPLETS = ["7dictzip", "7razf", "7gzinga", "7png", "7ciso", "7daxcr"]


def sub1(pat, repl, s, what, count=1, flags=re.S):
    out, n = re.subn(pat, repl, s, count=count, flags=flags)
    assert n >= 1, what
    return out


def edit(name, s):
    flushy = name in ("7dictzip", "7razf")         # chunks in full-flush form (applet/7dictzip.c:93-126, 7razf.c:126-160)
    enc = "hip_deflate"
    # 1. method -> func
    s = sub1(r"(\t*)\}else if\(method==DEFLATE_STORE\)\{\n(\t*)zlibbuf(->|\.)func = store_deflate;\n",
             lambda m: m.group(0) + "%s}else if(method==DEFLATE_HIP){\n%szlibbuf%sfunc = %s;\n" % (m.group(1), m.group(2), m.group(3), enc),
             s, "func ladder")
    # 2. error names
    s = sub1(r"(\t*)\}else if\(method==DEFLATE_STORE\)\{\n(\t*)fprintf\(stderr,\"store_deflate %d\\n\",zlibbuf(->|\.)ret\);\n",
             lambda m: m.group(0) + "%s}else if(method==DEFLATE_HIP){\n%sfprintf(stderr,\"%s %%d\\n\",zlibbuf%sret);\n" % (m.group(1), m.group(2), enc, m.group(3)),
             s, "error ladder")
    # 3. the level variables, the popt row, the option case
    s = sub1(r",kzip=0,store=0;", ",kzip=0,store=0,hip=0;", s, "vars")
    s = sub1(r"(\t*\{ \"store\",[^\n]*\n)",
             lambda m: m.group(1) + "\t\t{ \"hip\",     'G',         POPT_ARG_INT|POPT_ARGFLAG_OPTIONAL, NULL,    'G',       \"1-9 (default 1) hip (MI355X, libhipdeflate)\", \"level\" },\n",
             s, "popt row")
    s = sub1(r"(\t*)case 'T':\{\n(.*?\n)\1\}\n",
             lambda m: m.group(0) + "%scase 'G':{\n%s\tchar *arg=poptGetOptArg(optCon);\n%s\tif(arg)hip=strtol(arg,NULL,10),free(arg);\n%s\telse hip=1;\n%s\tbreak;\n%s}\n"
             % ((m.group(1),) * 6), s, "option case")
    # 4. exactly one method (7png.c has two of the three lines commented out: they are edited all the same)
    s = sub1(r"\+kzip\+store;", "+kzip+store+hip;", s, "level_sum")
    s = sub1(r"&&!kzip&&!store\)", "&&!kzip&&!store&&!hip)", s, "none check")
    s = sub1(r"\|\|kzip\|\|store\)\)", "||kzip||store||hip))", s, "mode check")
    s = sub1(r"\+\(level_sum==store\)!=1\)", "+(level_sum==store)+(level_sum==hip)!=1)", s, "one check")
    # 5. dispatch (behind the kzip branch, which every applet has)
    s = sub1(r"(\t*)\}else if\(kzip\)\{\n\1\tfprintf\(stderr,\"\(kzip\)\\n\"\);\n\1\tret=_compress\(([^\n]*?),kzip,DEFLATE_KZIP([^\n]*?)\);\n",
             lambda m: m.group(0) + "%s}else if(hip){\n%s\tfprintf(stderr,\"(hip)\\n\");\n%s\tret=_compress(%s,hip,DEFLATE_HIP%s);\n"
             % (m.group(1), m.group(1), m.group(1), m.group(2), m.group(3)), s, "dispatch")
    if flushy:
        # the chunk comes out of the kernel in full-flush form (HD_FRAME_RAW_FLUSH): no re-inflate with the patched zlib.
        # (7razf codes its LAST chunk with zlibutil_buffer_code -- a final block -- so the ladder keeps hip_deflate and
        # the switch to hip_deflate_flush sits here, where the reference makes a chunk non-final: INTEGRATION.md 3b)
        s = sub1(r"(static zlibutil_buffer \*zlibutil_buffer_full_flush\(zlibutil_buffer \*zlibbuf\)\{\n)(\tzlibutil_buffer_code\(zlibbuf\);\n)",
                 lambda m: m.group(1) + "\tif(zlibbuf->func==(void*)hip_deflate){ /* BFINAL=0 ... 00 00 ff ff straight from the kernel */\n"
                 "\t\tzlibbuf->func=hip_deflate_flush;zlibutil_buffer_code(zlibbuf);zlibbuf->func=hip_deflate;\n\t\treturn zlibbuf;\n\t}\n" + m.group(2),
                 s, "full flush")
        # ... and the reader of such chunks
        s = sub1(r"#if defined\(NOIGZIP\)\n(\t*// libdeflate_inflate cannot be used for Z_FULL_FLUSH stream\n\t*zlibbuf->func = zlib_inflate;\n)",
                 lambda m: "#if defined(USE_HIP_INFLATE)\n%szlibbuf->func = hip_inflate_flush;\n#elif defined(NOIGZIP)\n%s"
                 % (re.match(r"\t*", m.group(1)).group(0), m.group(1)), s, "chunk reader")
    return s


def main():
    for side in ("a", "b"):
        os.makedirs(os.path.join(WORK, side, "applet"), exist_ok=True)
    for name in APPLETS:
        src = open(os.path.join(REF, "applet", name + ".c"), encoding="latin-1").read()
        open(os.path.join(WORK, "a", "applet", name + ".c"), "w", encoding="latin-1").write(src)
        open(os.path.join(WORK, "b", "applet", name + ".c"), "w", encoding="latin-1").write(edit(name, src))
        p = subprocess.run(["diff", "-U1", "a/applet/%s.c" % name, "b/applet/%s.c" % name], cwd=WORK, capture_output=True)
        text = p.stdout.decode("latin-1")
        # (the header lines without diff's timestamps, as the hand-written hunks have them)
        text = re.sub(r"^(---|\+\+\+) (\S+)\t[^\n]*$", r"\1 \2", text, flags=re.M)
        sys.stdout.write("diff -U1 a/applet/%s.c b/applet/%s.c\n" % (name, name) + text)


if __name__ == "__main__":
    main()
