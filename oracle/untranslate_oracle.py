"""CPU oracle of `mugsy_profiles untranslate` (mugsy MAF over profile names -> MAF over the real genomes).
TEST INFRASTRUCTURE ONLY.

PARITY STATUS: RESTATED FROM SOURCE, NOT EXECUTED.  The reference is OCaml (lib/profiles/m_untranslate.ml and the OCaml
M_profile, which differs from the C++ one: lib/profiles/m_profile.ml:146-239) and cannot be built or run in this image
(SURVEY.md 8c).  Pinned only by the hand-derived fixture tests/golden/untranslate_handmade.* (derivation in
tests/test_untranslate.py)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

COMPLEMENT = {"A": "T", "a": "t", "T": "A", "t": "a", "C": "G", "c": "g", "G": "C", "g": "c"}  # m_untranslate.ml:15-24


class ProfileIdxOutOfRange(Exception):
    pass


class Profile:
    def __init__(self, major, minor, seq_name, rng, length, src_size, gaps, text):
        self.major, self.minor, self.seq_name = major, minor, seq_name
        self.range, self.length, self.src_size, self.gaps, self.text = rng, length, src_size, gaps, text


def read_profiles(text: str) -> List[Profile]:
    """lib/profiles/m_profile.ml:69-120 with ~lite:false, in file order."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    out, i = [], 0
    while i < len(lines):
        f = lines[i].split(" ")
        if len(f) != 7:
            raise ValueError("Error reading profile index file line " + lines[i])
        i += 1
        gaps = []
        while i < len(lines) and lines[i] != "0":
            a, b = lines[i].split(" ", 1)
            gaps.append((int(a), int(b)))
            i += 1
        i += 1  # the "0"
        if i >= len(lines):
            raise ValueError("Early end of file")
        out.append(Profile(f[0], f[1], f[2], (int(f[3]), int(f[4])), int(f[5]), int(f[6]), gaps, lines[i].strip()))
        i += 1
    return out


def seq_idx_of_profile_idx(p: Profile, pi: int) -> Optional[int]:
    """m_profile.ml:163-181"""
    if not pi < p.length + 1:
        raise ProfileIdxOutOfRange((pi, p.length))
    skipped = 0
    for gs, ge in p.gaps:
        if ge < pi:
            skipped += abs(gs - ge) + 1
        elif gs <= pi:
            return None
        else:
            break
    offset = pi - skipped - 1
    return p.range[0] + offset if p.range[0] <= p.range[1] else p.range[0] - offset


def subset_profile(p: Profile, s: int, e: int) -> Optional[Profile]:
    """m_profile.ml:189-239 (the OCaml one: no lower-bound check, None instead of an exception, p_length = |s-e|)."""
    if s > e:
        s, e = e, s
    if not (s < p.length + 1 and e < p.length + 1):
        raise ProfileIdxOutOfRange((s if s > p.length else e, p.length))
    gaps = []
    for gs, ge in p.gaps:
        a, b = (gs, ge) if gs < ge else (ge, gs)
        lo, hi = max(a, s), min(b, e)
        if hi - lo >= 0:
            gaps.append((lo, hi))
    if p.text != "":
        if s - 1 < 0 or (s - 1) + (e - s + 1) > len(p.text):
            raise IndexError("String.sub")
        text = p.text[s - 1:e]
    else:
        text = ""
    if len(gaps) == 1 and gaps[0] == (s, e):
        return None
    seq_s = seq_idx_of_profile_idx(p, gaps[0][1] + 1) if gaps and gaps[0][0] == s else seq_idx_of_profile_idx(p, s)
    seq_e = seq_idx_of_profile_idx(p, gaps[-1][0] - 1) if gaps and gaps[-1][1] == e else seq_idx_of_profile_idx(p, e)
    if seq_s is None or seq_e is None:
        return None
    return Profile(p.major, p.minor, p.seq_name, (seq_s, seq_e), abs(s - e), p.src_size, gaps, text)


def expand_text(p_text: str, text: str) -> str:
    """m_untranslate.ml:38-52"""
    out, k = [], 0
    for ch in text:
        if ch == "-":
            out.append("-")
        else:
            out.append(p_text[k])  # IndexError <-> Invalid_argument in the reference
            k += 1
    return "".join(out)


def of_maf(start, size, src_size, d):
    if d == "+":
        return (start + 1, start + size)
    if d == "-":
        return (src_size - start, src_size - start - (size - 1))
    raise ValueError("Invalid direction: " + d)


def untranslate(profile_files: List[str], in_maf: str) -> str:
    """m_untranslate.ml:127-221.  profile_files: texts of <dir>/profiles in -profile_paths_list order."""
    by_block: Dict[str, List[Profile]] = {}
    for t in profile_files:  # rows of a block end up in file order (two reversals, :26-36 and :153-166)
        for p in read_profiles(t):
            by_block.setdefault(p.major, []).append(p)
    lines = in_maf.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    out = ["##maf version=1 scoring=paramugsy"]
    for l in lines:
        if l.startswith("##maf "):
            continue
        if l == "" or l[0] == "#" or l.startswith("a score="):
            out.append(l)
        elif l.startswith("s "):
            tok = [t for t in l.replace("\t", " ").split(" ") if t != ""]
            if len(tok) != 7:
                raise ValueError("Unknown maf line: " + l)
            _, name, start, size, d, src_size, text = tok
            ov = of_maf(int(start), int(size), int(src_size), d)
            for p in by_block[name]:  # KeyError <-> Not_found
                sub = subset_profile(p, ov[0], ov[1])
                if sub is None:
                    continue
                p_fwd = p.range[0] <= p.range[1]
                if ov[0] <= ov[1]:  # get_real_range, :55-60
                    real, fwd = sub.range, p_fwd
                else:
                    real, fwd = (sub.range[1], sub.range[0]), not p_fwd
                length = abs(real[0] - real[1]) + 1
                if real[0] <= real[1]:  # get_start_size, :62-69
                    mstart = real[0] - 1
                else:
                    mstart = p.src_size - real[0]
                if p_fwd == fwd:
                    maf_text = expand_text(sub.text, text)
                else:
                    maf_text = "".join(COMPLEMENT.get(c, c) for c in expand_text(sub.text[::-1], text))
                out.append("s %s %d %d %s %d %s" % (p.seq_name, mstart, length, "+" if fwd else "-", p.src_size, maf_text))
        else:
            raise ValueError("Unknown line: " + l)
    return "\n".join(out) + "\n"
