"""Per-kernel register / scratch / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py dynamask_amd/csrc/backward.hip [name-filter] ; add --json FILE to save, --diff FILE to compare."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def usage(src):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
           '-I' + os.path.join(ROOT, 'dynamask_amd', 'csrc'), '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, name = {}, None
    for l in err.splitlines():
        m = re.search(r'Function Name: (\S+)', l)
        if m:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
            out[name] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+) \[-Rpass', l)
        if m and name:
            out[name][m.group(1).strip()] = int(m.group(2))
    return out

if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    u = usage(args[0])
    flt = args[1] if len(args) > 1 else ''
    if '--json' in sys.argv:
        json.dump(u, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
    old = json.load(open(sys.argv[sys.argv.index('--diff') + 1])) if '--diff' in sys.argv else None
    for k, v in u.items():
        if flt not in k:
            continue
        row = f"{k[:70]:70s} VGPR {v.get('VGPRs', -1):3d} AGPR {v.get('AGPRs', -1):3d} scratch {v.get('ScratchSize', -1):4d} LDS {v.get('LDS Size', -1):6d} occ {v.get('Occupancy', -1)}"
        if old is not None:
            o = old.get(k)
            if o == v:
                continue
            row += f"   <- was VGPR {o.get('VGPRs')} scratch {o.get('ScratchSize')} occ {o.get('Occupancy')}" if o else '   (new)'
        print(row)
