"""CPU oracles (test infrastructure).  See oracle/README.md.  Never imported by fish_speech_amd."""
