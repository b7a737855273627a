"""CPU oracle of `mugsy_profiles make` (MAF -> profiles + sequences.fasta).  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: RESTATED FROM SOURCE, NOT EXECUTED.  The reference for this stage is OCaml
(lib/profiles/m_make.ml, m_profile_stream.ml, m_profile.ml, m_range.ml) and no OCaml toolchain exists in the build
image, so this transcription is pinned only by the hand-derived fixture
tests/golden/make_handmade.* (expected bytes worked out by hand from the cited lines) -- not by running the
reference.  Plain Python, one character at a time, following the cited lines."""
from __future__ import annotations

from typing import List, Tuple


def gaps_of_text(text: str) -> List[Tuple[int, int]]:
    """lib/profiles/m_profile.ml:29-47: 1-based inclusive runs of '-'."""
    out = []
    i, n = 0, len(text)
    while i < n:
        if text[i] == "-":
            j = i + 1
            while j < n and text[j] == "-":
                j += 1
            out.append((i + 1, j))
            i = j
        else:
            i += 1
    return out


def combine_text(a: str, b: str) -> str:
    """lib/profiles/m_make.ml:15-28."""
    assert len(a) == len(b)
    out = []
    for x, y in zip(a, b):
        if x == y:
            out.append(x)
        elif x != "-" and y != "-":
            out.append("N")
        elif x != "-":
            out.append(x)
        else:
            out.append(y)
    return "".join(out)


def of_maf(start: int, size: int, src_size: int, direction: str) -> Tuple[int, int]:
    """lib/profiles/m_range.ml:60-65."""
    if direction == "+":
        return (start + 1, start + size)
    if direction == "-":
        return (src_size - start, src_size - start - (size - 1))
    raise ValueError("Invalid direction: " + direction)  # m_profile_stream.ml:11-14


def make(maf_text: str, basename: str) -> Tuple[str, str]:
    """-> (profiles file text, sequences.fasta text).  m_profile_stream.ml:23-74, m_profile.ml:122-135, m_make.ml:31-62."""
    lines = maf_text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()  # read_file_lines yields no line after a final newline
    pos = 0
    profiles: List[str] = []
    fasta: List[str] = []
    count = 0
    while True:
        while pos < len(lines) and not lines[pos].startswith("a score="):  # drop_until_score
            pos += 1
        if pos >= len(lines):
            break
        pos += 1
        major = "%s.%s_%04d" % (basename, basename, count)
        idx = 0
        cons = None
        while True:  # stream_profiles
            if pos >= len(lines):
                if idx == 0:
                    raise ValueError("Expected alignment, did not get")
                break
            s = lines[pos]
            pos += 1
            if len(s) == 0:
                break
            if s.startswith("s "):
                tok = [t for t in s.replace("\t", " ").split(" ") if t != ""]
                if len(tok) != 7:
                    raise ValueError("Unknown maf line: " + s)
                _, name, start, size, d, src_size, text = tok
                rs, re = of_maf(int(start), int(size), int(src_size), d)
                profiles.append("%s %d %s %d %d %d %d\n" % (major, idx, name, rs, re, len(text), int(src_size)))
                for gs, ge in gaps_of_text(text):
                    profiles.append("%d %d\n" % (gs, ge))
                profiles.append("0\n")
                profiles.append(text + "\n")
                cons = text if cons is None else combine_text(cons, text)
                idx += 1
            elif s[0] == "#":
                continue
            else:
                raise ValueError("Unknown line")
        if cons is not None:
            fasta.append(">%s\n%s\n\n" % (major, cons))
        count += 1
    return "".join(profiles), "".join(fasta)
