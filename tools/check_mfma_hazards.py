"""Static check of gfx950 assembly: is an MFMA result read too early?

gfx950 does not interlock a VALU / LDS / memory instruction that touches the destination registers of an MFMA
still in the matrix pipe: software has to put `passes + 4` wait states between the two for the fp16 / bf16 / fp8 /
int8 shapes (LLVM's GFX940_XDL_N_PassWriteVgprVALURawWaitStates with the gfx950 increment) and `passes + 2` for the
f32-operand shapes (..._SMFMA_...); an instruction is one wait state, `s_nop N` is N + 1.  The compiler's hazard recognizer normally does that, but it was caught emitting 3-5 wait states
instead of 12 in front of the `max` tree of the filter kernel when the survivor append was a peeled side branch
(profiles/r05_scan16f_peel.txt): the kernel then read a stale accumulator now and then and lost
survivors of its last query group; the same fault sat, harmless so far, in the direct-store f32 DotInteraction kernel.  This walks every kernel of a `.s` file in layout order (the fall-through
path; behind an unconditional branch the count restarts) and, from every branch it meets while an MFMA is still in
the pipe, the TAKEN path as well (three branches deep) -- side blocks and loop back edges -- and reports each read
that comes too early.

  python tools/check_mfma_hazards.py file.s [kernel-name-substring]
"""
import re
import sys

# passes of the matrix pipe per instruction (4 cycles each), gfx950
_PASSES = [
    (r"v_mfma_f32_32x32x16_(f16|bf16)", 8),
    (r"v_mfma_f32_16x16x32_(f16|bf16)", 4),
    (r"v_mfma_f32_32x32x8_?(f16|bf16_1k|xf32)", 16),
    (r"v_mfma_f32_16x16x16_?(f16|bf16_1k)", 8),
    (r"v_mfma_f32_32x32x2_?f32", 16),
    (r"v_mfma_f32_16x16x4_?f32", 8),
    (r"v_mfma_f32_32x32x4_?(f16|bf16)", 16),
    (r"v_mfma_f32_4x4x", 2),
    (r"v_mfma_f32_32x32x16_(fp8|bf8)", 8),
    (r"v_mfma_f32_16x16x32_(fp8|bf8)", 4),
    (r"v_mfma_f32_32x32x64_f8f6f4", 16),
    (r"v_mfma_f32_16x16x128_f8f6f4", 8),
    (r"v_mfma_i32_32x32x32_i8", 8),
    (r"v_mfma_i32_16x16x64_i8", 4),
    (r"v_mfma_f64_16x16x4_?f64", 8),
]
_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def _passes(mnemonic):
  for pat, n in _PASSES:
    if re.match(pat, mnemonic):
      return n
  return 16   # unknown shape: the longest


def _regs(text):
  out = []
  for m in _REG.finditer(text):
    lo = int(m.group(2) if m.group(2) is not None else m.group(3))
    hi = int(m.group(2) if m.group(2) is not None else m.group(4))
    out.append((m.group(1), lo, hi))
  return out


def kernels(asm):
  """(name, [(line_no, text)]) per kernel of an assembly listing."""
  out, name, body = [], None, []
  for no, line in enumerate(asm.splitlines(), 1):
    m = re.match(r"^(_Z\w+):", line)
    if m:
      name, body = m.group(1), []
      continue
    if name is None:
      continue
    if line.lstrip().startswith(".amdhsa_kernel") or line.lstrip().startswith(".section"):
      out.append((name, body))
      name = None
      continue
    body.append((no, line))
  return out


_BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\.?\w+)")


def _parse(body):
  """[(line_no, op, operands, text)] of the instructions and {label: index of the instruction behind it}."""
  ins, labels = [], {}
  for no, raw in body:
    text = raw.split(";")[0].strip()
    if not text or text.startswith("."):
      if text.endswith(":"):
        labels[text[:-1]] = len(ins)
      continue
    if text.endswith(":"):
      labels[text[:-1]] = len(ins)
      continue
    parts = text.split(None, 1)
    ins.append((no, parts[0], parts[1] if len(parts) > 1 else "", text))
  return ins, labels


def _required(op):
  n = _passes(op)
  xdl = not re.search(r"x\d+_?f32$|_f64$", op.split("_e64")[0])     # f32 / f64 operands: not the XDL pipe's rule
  return (n + 3 + (1 if n != 2 else 0)) if xdl else n + 2


def _walk(ins, labels, start, pending, bad, seen, depth):
  """Follows the fall-through path from instruction `start` with the MFMAs in `pending` still in the pipe; at every
  branch the TAKEN path is followed as well (a copy of the pending set, at most three branches deep) until the pipe
  has delivered everything.  `start == 0` is the whole-kernel walk, which also picks up new MFMAs."""
  top = depth == 0
  i = start
  while i < len(ins):
    if not top and not pending:
      return
    no, op, operands, text = ins[i]
    m = _BRANCH.match(text)
    if m:
      if pending and depth < 3 and m.group(2) in labels:
        key = (labels[m.group(2)], tuple((p[0], p[1], p[2], p[3]) for p in pending))
        if key not in seen:
          seen.add(key)
          taken = [[p[0], p[1], p[2], p[3] - 1, p[4]] for p in pending if p[3] - 1 > 0]   # the branch is one wait state
          _walk(ins, labels, labels[m.group(2)], taken, bad, seen, depth + 1)
      if m.group(1) == "s_branch":
        if not top:
          return
        pending = []
        i += 1
        continue
    if op in ("s_endpgm", "s_setpc_b64"):
      if not top:
        return
      pending = []
      i += 1
      continue
    is_mfma = op.startswith("v_mfma") or op.startswith("v_smfmac")
    if not is_mfma and pending and (op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_"))):
      for f, lo, hi in _regs(operands):
        for p in pending:
          if p[0] == f and lo <= p[2] and hi >= p[1] and p[3] > 0:
            rec = (no, text, f"{f}[{lo}:{hi}]", p[3], p[4])
            if rec not in bad:
              bad.append(rec)
            p[3] = 0   # one report per MFMA and path
    states = int(operands.strip(), 0) + 1 if op == "s_nop" else 1
    for p in pending:
      p[3] -= states
    pending = [p for p in pending if p[3] > 0]
    if is_mfma:
      regs = _regs(operands)
      if regs:
        f, lo, hi = regs[0]
        pending = [p for p in pending if not (p[0] == f and p[1] == lo and p[2] == hi)]   # the chain's next link
        if top:
          pending.append([f, lo, hi, _required(op), no])
    i += 1


def short_reads(body):
  """[(line_no, instruction, register, missing wait states, line of the MFMA)] of one kernel body."""
  ins, labels = _parse(body)
  bad = []
  _walk(ins, labels, 0, [], bad, set(), 0)
  return bad


def check(asm, only=None):
  """{kernel: [short reads]} over the kernels of a listing that contain MFMAs."""
  out = {}
  for name, body in kernels(asm):
    if only and only not in name:
      continue
    if not any("v_mfma" in l or "v_smfmac" in l for _, l in body):
      continue
    out[name] = short_reads(body)
  return out


if __name__ == "__main__":
  res = check(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else None)
  n_bad = 0
  for name, bad in res.items():
    print(f"{name}: {len(bad)} early read(s)")
    for no, text, reg, missing, src in bad[:12]:
      print(f"  line {no}: {text}   <- {reg} of the MFMA at line {src}, {missing} wait state(s) short")
    n_bad += len(bad)
  sys.exit(1 if n_bad else 0)
