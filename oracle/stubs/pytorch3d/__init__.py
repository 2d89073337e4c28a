"""TEST INFRASTRUCTURE ONLY.  Stand-in for the third-party `pytorch3d` package (pinned 0.7.5 by the
reference's README.md:29) which is absent from this image.  Only `pytorch3d.transforms` is provided and only
the four functions the reference's hot path calls.  Used solely by tests/golden/make_golden.py."""
