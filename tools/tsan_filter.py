import re, sys
txt = open(sys.argv[1]).read()
reports = [r for r in txt.split("==================") if "WARNING: ThreadSanitizer" in r]
hits = []
for r in reports:
    blocks = re.split(r"\n(?=  (?:Write|Read|Previous|Atomic|Location|Mutex|Thread) )", r)
    acc = [b for b in blocks if re.match(r"  (Write|Read|Previous write|Previous read|Atomic|Previous atomic)", b)]
    ours = 0
    for b in acc:
        frames = re.findall(r"#(\d+) (.*)", b)
        top = [f for n, f in frames if int(n) <= 1]
        if any("dtv-utils_amd/" in f for f in top):
            ours += 1
    if ours:
        hits.append((ours, r))
print("  reports:", len(reports), "- with one of the two accesses in our code (frames #0/#1):", len(hits))
for ours, r in hits[:5]:
    print("  ---")
    print("\n".join("    " + l[:240] for l in r.strip().splitlines()[:28]))
