# First contact of the Elixir host with a real BEAM, in ONE command and without a project checkout:
#
#     make -C nif                      # priv/nxsig_nif.so against the installed OTP's erl_nif.h (after `make -C nif check-header`)
#     elixir elixir/smoke.exs          # needs OTP + Elixir >= 1.15, a GPU, and (first run only) network for Mix.install(:nx)
#
# It loads the dirty NIF (NXSIG_NIF_PATH, default elixir/priv/nxsig_nif.so), compiles the host modules of elixir/lib, runs the
# reference's stft doctest (elixir-nx/nx_signal v0.3.0 lib/nx_signal.ex:46-65) and one device-resident stft -> istft round trip,
# and prints ONE line: "nxsig smoke PASS ..." or "nxsig smoke FAIL ...".  Exit status 0 / 1.  Not run in the build image (no BEAM);
# tests/test_elixir_sources.py checks its structure and that everything it calls exists in elixir/lib.
Mix.install([{:nx, "~> 0.11"}])

root = Path.dirname(__ENV__.file)
System.put_env("NXSIG_NIF_PATH", System.get_env("NXSIG_NIF_PATH") || Path.join([root, "priv", "nxsig_nif.so"]))

result =
  try do
    unless File.exists?(System.get_env("NXSIG_NIF_PATH")) do
      raise "#{System.get_env("NXSIG_NIF_PATH")} not found: run `make -C nif` first (or point NXSIG_NIF_PATH at nxsig_nif.so)"
    end

    # the host modules, in whatever order their struct / macro dependencies need (the parallel compiler sorts that out)
    {:ok, _modules, _warnings} = Kernel.ParallelCompiler.require(Path.wildcard(Path.join([root, "lib", "**", "*.ex"])), return_diagnostics: false)

    sig = NxSignalAMD
    ctx = sig.context(0)

    # 1. the reference's own stft doctest: values, names, times, frequencies
    {z, t, f} = sig.stft(Nx.iota({4}), NxSignalAMD.Windows.rectangular(2), overlap_length: 1, fft_length: 2, sampling_rate: 400)
    true = Nx.shape(z) == {3, 2} and Nx.names(z) == [:frames, :frequencies]
    true = Nx.to_flat_list(Nx.real(z)) == [1.0, -1.0, 3.0, -1.0, 5.0, -1.0]
    true = Nx.to_flat_list(Nx.imag(z)) == [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    true = Nx.to_flat_list(f) == [0.0, 200.0]
    true = Nx.to_flat_list(t) == Enum.map([0.0025, 0.005, 0.0075], &Nx.to_number(Nx.tensor(&1, type: :f32)))

    # 2. one second of audio through the tuned kernels, device-resident: stft -> istft reproduces the interior samples
    x = Nx.iota({48_000}, type: :f32) |> Nx.multiply(0.01) |> Nx.sin()
    w = NxSignalAMD.Windows.hann(1024)
    opts = [overlap_length: 768, fft_length: 1024, sampling_rate: 48_000]
    xd = NxSignalAMD.DeviceTensor.to_device(x)
    {zd, _t, _f} = sig.stft(xd, w, opts)
    family_fwd = sig.last_dispatch(ctx)
    yd = sig.istft(zd, w, opts)
    family_inv = sig.last_dispatch(ctx)
    y = yd |> NxSignalAMD.DeviceTensor.from_device() |> Nx.real()
    n = elem(Nx.shape(y), 0)
    err = Nx.to_number(Nx.reduce_max(Nx.abs(Nx.subtract(y[1024..(n - 1025)], x[1024..(n - 1025)]))))
    true = err < 1.0e-5
    true = String.starts_with?(family_fwd, "stft.pair") and String.starts_with?(family_inv, "istft.wave")

    {:ok, "doctest exact, round trip max err #{:erlang.float_to_binary(err, scientific: 2)}, kernels #{family_fwd} / #{family_inv}"}
  rescue
    e -> {:error, Exception.format(:error, e, __STACKTRACE__) |> String.split("\n") |> Enum.take(3) |> Enum.join(" | ")}
  catch
    kind, reason -> {:error, "#{kind}: #{inspect(reason)}"}
  end

case result do
  {:ok, msg} ->
    IO.puts("nxsig smoke PASS: #{msg}")

  {:error, msg} ->
    IO.puts("nxsig smoke FAIL: #{msg}")
    System.halt(1)
end
