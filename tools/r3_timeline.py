"""One steady-state iteration of a joint fit from a rocprofv3 kernel trace: python tools/r3_timeline.py <dir with *_kernel_trace.csv> [marker]
marker: substring of the kernel that ends an iteration (default: the reduction + update kernel)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else 'update_kernel'
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[i0]['End_Timestamp'])
for r in rows[i0:i1 + 1]:
    name = r['Kernel_Name']
    short = name.split('(')[0][-48:]
    if 'joint_epoch_kernel' in name:
        short = 'epoch phase' + name.split('>')[-2][-3:]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r['Queue_Id']} "
          f"grid {r['Grid_Size_X']:>6}x{r['Grid_Size_Y']:<3} wg {r['Workgroup_Size_X']:>4} vgpr {r['VGPR_Count']:>3} lds {r['LDS_Block_Size']:>6}  {short}")
