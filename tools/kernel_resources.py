#!/usr/bin/env python3
"""Register / scratch / static-LDS budget of the kernels in the built objects (csrc/build/*.o): the table of DESIGN.md section 4.
usage: kernel_resources.py [substring ...]   (no GPU needed: reads the gfx950 code objects with llvm-readelf)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
want = sys.argv[1:]
with tempfile.TemporaryDirectory() as tmp:
    for obj in sorted(os.listdir(os.path.join(ROOT, 'pynucleus_amd', 'csrc', 'build'))):
        if not obj.endswith('.o'):
            continue
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'k.co')
        subprocess.run([LLVM+'/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', os.path.join(ROOT, 'pynucleus_amd', 'csrc', 'build', obj), fat], check=True)
        r = subprocess.run([LLVM+'/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input='+fat, '--output='+co, '--unbundle'], capture_output=True)
        if r.returncode or not os.path.exists(co):
            continue
        txt = subprocess.run([LLVM+'/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
        rows = []
        for b in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
            b = '.agpr_count'+b
            g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, b) or [None, None])[1]
            rows.append([g('name'), g('vgpr_count'), g('agpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')])
        names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
        for d, r in zip(names, rows):
            short = re.sub(r'\(.*', '', d.replace('(anonymous namespace)::', '')).replace('void ', '')
            if want and not any(w in short for w in want):
                continue
            print('{:14s} {:60s} vgpr {:>3s} agpr {:>3s} scratch {:>4s} B  static LDS {:>6s} B'.format(obj, short[:60], r[1], r[2], r[3], r[4]))
        os.remove(co)
