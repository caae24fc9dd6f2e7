"""ORACLE — TEST INFRASTRUCTURE ONLY.  See oracle/README.md.  Never imported by rnntransducer_amd/."""
