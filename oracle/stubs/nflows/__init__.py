"""TEST INFRASTRUCTURE ONLY.  Stand-in for the third-party `nflows` package (reference README.md:33, unpinned)."""
