"""Selected counters of selected dispatches from a tools/pmc_run.sh dispatch table.
    python tools/pmc_pick.py <dispatches.txt> <kernel substring> [...]"""
import sys
KEYS = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
        'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INST_LEVEL_SMEM',
        'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INST_LEVEL_VMEM', 'SQ_INSTS_LDS', 'SQ_WAIT_INST_LDS', 'SQ_IFETCH', 'SQ_IFETCH_LEVEL',
        'SQC_ICACHE_REQ', 'SQC_ICACHE_MISSES', 'SQC_DCACHE_REQ', 'SQC_DCACHE_MISSES', 'SQ_THREAD_CYCLES_VALU', 'SQ_INSTS_VALU_INT32',
        'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_TRANS_F32', 'GRBM_GUI_ACTIVE', 'FETCH_SIZE', 'WRITE_SIZE']
want = sys.argv[2:]
hdr = None
for l in open(sys.argv[1]).read().splitlines():
    if l.startswith('idx kernel'):
        hdr = l.split()[2:]
    elif hdr and l and l[0].isdigit():
        parts = l.split()
        n = len(hdr)
        name = ' '.join(parts[1:-n])
        if want and not any(w in name for w in want):
            continue
        d = dict(zip(hdr, [float(v) for v in parts[-n:]]))
        print(parts[0], name)
        print('   ' + '  '.join(f"{k.replace('SQ_', '').replace('INSTS_', 'I_')}={d[k]:.4g}" for k in KEYS if k in d))
