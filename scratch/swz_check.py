# ds_read_b128 bank-conflict check for [row][8 slots x 16 B] images (128-B rows), slot' = kslot ^ f(row)
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(rowfn, f, ks):
    worst = 1
    for g in groups:
        seen = {}
        for l in g:
            fr, fg = l & 15, l >> 4
            row = rowfn(fr)
            addr = row*128 + ((fg + 4*ks) ^ f(row))*16
            b = (addr // 16) % 16
            seen.setdefault(b, set()).add(addr)
        worst = max(worst, max(len(v) for v in seen.values()))
    return worst
f1 = lambda r: (r >> 1) & 7
for name, f in [('(r>>1)&7', f1), ('r&7', lambda r: r & 7), ('(r>>1)&7 ^ ((r>>4)&1)', lambda r: ((r>>1)&7))]:
    for stride in (1, 2):
        res = {}
        for base in range(0, 64):
            for ks in (0, 1):
                w = conflicts(lambda fr: base + stride*fr, f, ks)
                res[w] = res.get(w, 0) + 1
        print(name, 'stride', stride, res)
