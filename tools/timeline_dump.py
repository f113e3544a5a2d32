import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, queue_id, stream_id, start, end, grid_x, grid_y from kernels order by start").fetchall()
# keep the second half (after warm-up): find the last gram_kernel with the largest grid
gi = [i for i, r in enumerate(rows) if 'gram_kernel' in r[0]]
rows = rows[gi[-1]:]
t0 = rows[0][3]
with open(sys.argv[2], 'w') as f:
    f.write("name,queue,stream,start_us,end_us,dur_us,grid_x,grid_y\n")
    for n, q, s, a, b, gx, gy in rows:
        import re
        m = re.search(r'gemm_nt_kernel<(\d), (\d), (\d)', n)
        m2 = re.search(r'gemm_nt128(_persist)?_kernel<(\d), (\d)', n)
        short = (f'gemm{m.group(1)}_{m.group(2)}{m.group(3)}' if m else f'big{m2.group(2)}_{m2.group(3)}{"p" if m2.group(1) else ""}' if m2
                 else 'potf2' if 'potf2' in n else n.split('(')[0].split('::')[-1][:20].replace(',', ';'))
        f.write(f"{short},{q},{s},{(a - t0) / 1e3:.1f},{(b - t0) / 1e3:.1f},{(b - a) / 1e3:.1f},{gx},{gy}\n")
print(len(rows), "kernels; span", (rows[-1][4] - t0) / 1e6, "ms")
