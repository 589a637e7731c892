"""Entry point with the reference's dispatch (ref: run_me.py:7-32): `python run_me.py icrl <flags>` / `python run_me.py cpg <flags>`."""
import sys

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("icrl", "cpg"):
        raise SystemExit("usage: python run_me.py {icrl,cpg} <flags>   (gail / run_policy / random_agent are outside the hot path)")
    if sys.argv[1] == "icrl":
        from icrl_amd.icrl import main
    else:
        from icrl_amd.cpg import main
    main(sys.argv[1:])
