#!/usr/bin/env python3
"""Says which sources a piece of evidence under profiles/ was taken on.

    python tools/stamp.py write        build container, before a gpurun call: gpurun_stamp.json <- git HEAD + hashes
    python tools/stamp.py line         one-line stamp of the tree this runs in (header of text evidence)
    python tools/stamp.py json         the same as a JSON object
    python tools/stamp.py embed [--from STAMP.json] FILE…
                                       add the stamp (of this tree, or the one a GPU-box run left in STAMP.json) to evidence
                                       files in place: a "stamp" key in *.json, a first "# stamp: …" line in anything else
    python tools/stamp.py check        every profiles/r05_* file's stamp against this tree (exit 1 when one is stale)

The GPU box has no .git: the commit travels in gpurun_stamp.json (git-ignored, shipped by gpurun), and is trusted only
while the hash of the kernel sources it was written for -- csrc_sha: every file of visual-odometry_amd/csrc that the
library is built from -- still equals that of the tree at hand.  bench.py quotes committed counter summaries with their
stamp and calls them stale when their csrc_sha is not the running tree's."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visual-odometry_amd", "csrc")
LIB = os.path.join(ROOT, "visual-odometry_amd", "libvo_hip.so")
STAMP_FILE = os.path.join(ROOT, "gpurun_stamp.json")


def csrc_sha(root=ROOT):
    """sha256 over the names and contents of the library's sources (csrc/*.hip, *.h, Makefile; include/vo_hip.h)"""
    h = hashlib.sha256()
    d = os.path.join(root, "visual-odometry_amd", "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")])
    files.append(os.path.join(root, "include", "vo_hip.h"))
    for f in files:
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def lib_sha(path=LIB):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def git_head(root=ROOT):
    try:
        r = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if r.returncode != 0:
            return None, None
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "visual-odometry_amd/csrc", "include/vo_hip.h"],
                               capture_output=True, text=True, timeout=10).stdout.strip() != ""
        return r.stdout.strip()[:12], dirty
    except (OSError, subprocess.SubprocessError):
        return None, None


def current():
    """the stamp of the tree this runs in"""
    sha = csrc_sha()
    commit, dirty = git_head()
    if commit is None:                       # GPU box: the commit comes with the snapshot
        try:
            w = json.load(open(STAMP_FILE))
            commit = w["commit"] if w.get("csrc_sha") == sha else "unknown (sources differ from gpurun_stamp.json)"
            dirty = w.get("csrc_dirty", False) if w.get("csrc_sha") == sha else None
        except (OSError, ValueError, KeyError):
            commit = "unknown (no gpurun_stamp.json)"
    return {"commit": commit, "csrc_dirty": dirty, "csrc_sha": sha, "lib_sha256": lib_sha()}


def line(st=None):
    st = st or current()
    return "stamp: commit=%s%s csrc_sha=%s lib_sha256=%s" % (st["commit"], "+uncommitted-csrc" if st.get("csrc_dirty") else "",
                                                             st["csrc_sha"], st["lib_sha256"])


def read_stamp(path):
    """the stamp embedded in an evidence file, or None"""
    try:
        if path.endswith(".json"):
            d = json.load(open(path))
            return d.get("stamp") or d.get("evidence_stamp")
        with open(path, errors="replace") as f:
            for _ in range(3):
                l = f.readline()
                if "stamp: commit=" in l:
                    kv = dict(x.split("=", 1) for x in l.split("stamp: ", 1)[1].split())
                    return {"commit": kv.get("commit"), "csrc_sha": kv.get("csrc_sha"), "lib_sha256": kv.get("lib_sha256")}
    except (OSError, ValueError):
        pass
    return None


def status(path, sha=None):
    """'current' / 'stale' / 'unstamped' for an evidence file against the running tree's kernel sources"""
    st = read_stamp(path)
    if not st or not st.get("csrc_sha"):
        # a file that cannot carry a stamp (csv): the round's list speaks for it (profiles/rNN_stamp.json)
        side = os.path.join(os.path.dirname(path), os.path.basename(path)[:3] + "_stamp.json")
        try:
            d = json.load(open(side))
            if os.path.basename(path) in d.get("files", []):
                st = d["stamp"]
        except (OSError, ValueError, KeyError):
            pass
    if not st or not st.get("csrc_sha"):
        return "unstamped", st
    return ("current" if st["csrc_sha"] == (sha or csrc_sha()) else "stale"), st


def embed(paths, st=None):
    st = st or current()
    for p in paths:
        if p.endswith(".json"):
            txt = open(p).read()
            body = [l for l in txt.splitlines() if l.startswith("{")]
            d = json.loads(body[-1] if len(body) and not txt.lstrip().startswith("{\n") else txt)
            d["stamp"] = st
            json.dump(d, open(p, "w"), indent=1 if "\n" in txt.strip() else None)
        else:
            txt = open(p, errors="replace").read()
            if "stamp: commit=" in txt[:400]:
                continue
            open(p, "w").write("# " + line(st) + "\n" + txt)


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "line"
    if cmd == "write":
        st = current()
        json.dump(st, open(STAMP_FILE, "w"))
        print(line(st))
    elif cmd == "line":
        print(line())
    elif cmd == "json":
        print(json.dumps(current()))
    elif cmd == "embed":
        args = sys.argv[2:]
        st = None
        if args and args[0] == "--from":          # the stamp of the run that produced the files (written on the GPU box)
            st = json.load(open(args[1]))
            args = args[2:]
        embed(args, st)
    elif cmd == "check":
        sha, bad = csrc_sha(), 0
        for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_*"))):
            s, st = status(p, sha)
            print("%-10s %-48s %s" % (s, os.path.basename(p), (st or {}).get("commit", "")))
            bad += s == "stale"
        sys.exit(1 if bad else 0)
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
