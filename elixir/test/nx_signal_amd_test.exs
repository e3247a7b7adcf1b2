defmodule NxSignalAMDTest do
  # Values come from the reference's own doctests (elixir-nx/nx_signal v0.3.0: lib/nx_signal.ex:46-65, :182-204, :545-554,
  # :656-681; lib/nx_signal/windows.ex:266-275) — the same literals tests/golden/reference_vectors.json holds for the Python
  # mirror and the NIF harness.  Not run in the build image (no BEAM); `cd elixir && mix test` on a machine with OTP + a GPU.
  use ExUnit.Case, async: false

  alias NxSignalAMD, as: Sig

  test "stft doctest: rectangular window of 2, hop 1, fs 400" do
    {z, t, f} = Sig.stft(Nx.iota({4}), Sig.Windows.rectangular(2), overlap_length: 1, fft_length: 2, sampling_rate: 400)
    assert Nx.shape(z) == {3, 2}
    assert Nx.names(z) == [:frames, :frequencies]
    assert Nx.to_flat_list(Nx.real(z)) == [1.0, -1.0, 3.0, -1.0, 5.0, -1.0]
    assert Nx.to_flat_list(t) == [0.0025, 0.005, 0.0075] |> Enum.map(&Nx.to_number(Nx.tensor(&1, type: :f32)))
    assert Nx.to_flat_list(f) == [0.0, 200.0]
  end

  test "as_windowed keeps integer tensors exact" do
    out = Sig.as_windowed(Nx.iota({10}), window_length: 4)
    assert Nx.type(out) == {:s, 64}
    assert Nx.to_list(out) == for(i <- 0..6, do: Enum.to_list(i..(i + 3)))
  end

  test "overlap_and_add doctest" do
    assert Nx.to_flat_list(Sig.overlap_and_add(Nx.iota({3, 4}), overlap_length: 0)) == Enum.to_list(0..11)
    assert Nx.to_flat_list(Sig.overlap_and_add(Nx.iota({3, 4}), overlap_length: 3)) == [0, 5, 15, 18, 17, 11]

    assert_raise ArgumentError, ~r/overlap_length must be a number less than the window size 4, got: 4/, fn ->
      Sig.overlap_and_add(Nx.iota({3, 4}), overlap_length: 4)
    end
  end

  test "hann window is bit-identical to the reference's doctest" do
    assert Nx.to_flat_list(Sig.Windows.hann(5, is_periodic: false)) == [0.0, 0.5, 1.0, 0.5, 0.0]
  end

  test "stft |> istft round trip stays on the device" do
    x = Nx.iota({4096}, type: :f32) |> Nx.sin()
    w = Sig.Windows.hann(1024)
    opts = [overlap_length: 768, fft_length: 1024, sampling_rate: 48_000]
    xd = Sig.DeviceTensor.to_device(x)
    {zd, _t, _f} = Sig.stft(xd, w, opts)
    assert %Sig.DeviceTensor{shape: {13, 1024}, type: {:c, 64}} = zd
    y = zd |> Sig.istft(w, opts) |> Sig.DeviceTensor.from_device() |> Nx.real()
    assert Nx.to_number(Nx.reduce_max(Nx.abs(Nx.subtract(y[1024..3071], x[1024..3071])))) < 1.0e-5
  end

  test "complex samples (c64 IQ data) are framed, windowed and transformed like the reference does" do
    # lib/nx_signal.ex:94-102 on a complex tensor: stft is linear, so stft(a + i b) == stft(a) + i stft(b) to fp32 round-off, and a
    # complex exponential at bin 3 of 16 puts its energy into bin 3 only (a real signal's spectrum would mirror it into bin 13)
    n = Nx.iota({64}, type: :f32)
    a = Nx.cos(Nx.multiply(n, 2 * :math.pi() * 3 / 16))
    b = Nx.sin(Nx.multiply(n, 2 * :math.pi() * 3 / 16))
    x = Nx.complex(a, b)
    w = Sig.Windows.rectangular(16)
    opts = [overlap_length: 0, fft_length: 16, sampling_rate: 16]
    {z, _t, _f} = Sig.stft(x, w, opts)
    assert Nx.type(z) == {:c, 64} and Nx.shape(z) == {4, 16}
    mag = Nx.abs(z[0])
    assert Nx.to_number(mag[3]) > 15.9
    assert Nx.to_number(mag[13]) < 1.0e-4
    {za, _, _} = Sig.stft(a, w, opts)
    {zb, _, _} = Sig.stft(b, w, opts)
    sum = Nx.add(za, Nx.multiply(zb, Nx.complex(0.0, 1.0)))
    assert Nx.to_number(Nx.reduce_max(Nx.abs(Nx.subtract(z, sum)))) < 1.0e-4
    {zd, _, _} = Sig.stft(Sig.DeviceTensor.to_device(x), w, opts)
    assert Sig.DeviceTensor.from_device(zd) == z
    # c128 samples, and c64 samples under an f64 window, compute in c128 (the product of :101 promotes)
    {z128, _, _} = Sig.stft(Nx.as_type(x, :c128), Sig.Windows.rectangular(16, type: :f64), opts)
    assert Nx.type(z128) == {:c, 128}
    assert Nx.to_number(Nx.reduce_max(Nx.abs(Nx.subtract(z128, Nx.as_type(z, :c128))))) < 1.0e-4
    {zmix, _, _} = Sig.stft(x, Sig.Windows.rectangular(16, type: :f64), opts)
    assert Nx.type(zmix) == {:c, 128}
  end

  test "vectorized (multichannel) inputs keep their vectorized axes" do
    x = Nx.iota({3, 2048}, type: :f32) |> Nx.vectorize(:channel)
    {z, _t, _f} = Sig.stft(x, Sig.Windows.hann(256), overlap_length: 192, sampling_rate: 8_000)
    assert z.vectorized_axes == [channel: 3]
    assert Nx.shape(z) == {29, 256}
  end

  test "invalid options raise ArgumentError like the reference" do
    assert_raise ArgumentError, ~r/invalid :scaling/, fn -> Sig.stft(Nx.iota({16}), Sig.Windows.hann(4), scaling: :eggs) end
    assert_raise ArgumentError, ~r/invalid padding mode/, fn -> Sig.stft(Nx.iota({16}), Sig.Windows.hann(4), window_padding: :zeros) end
  end

  test "convolve/3 defaults to the direct method and is exact on integer data" do
    assert Sig.Convolution.convolve(Nx.tensor([3, 4, 5, 6, 5, 4]), Nx.tensor([1, 2, 3])) ==
             Nx.tensor([3, 10, 22, 28, 32, 32, 23, 12], type: :f32)

    assert Sig.Convolution.convolve(Nx.tensor([3, 4, 5]), Nx.tensor([1, 2, 3, 4]), mode: :same) == Nx.tensor([10, 22, 34], type: :f32)
    assert Sig.Convolution.correlate(Nx.tensor([1, 2, 3]), Nx.tensor([3, 4, 5])) == Nx.tensor([5.0, 14.0, 26.0, 18.0, 9.0])
    assert Sig.Convolution.convolve(Nx.tensor(1289), Nx.tensor(4567)) == Nx.tensor(5_886_863.0)
    assert_raise ArgumentError, fn -> Sig.Convolution.convolve(Nx.tensor([1]), Nx.tensor(2)) end
  end

  test "istft_filtered equals multiply-then-istft" do
    z = Nx.iota({2, 9, 1024}, type: :f32) |> Nx.sin() |> Nx.as_type(:c64)
    h = Nx.iota({1024}, type: :f32) |> Nx.cos() |> Nx.as_type(:c64)
    w = Sig.Windows.hann(1024)
    opts = [overlap_length: 768, sampling_rate: 48_000]
    assert Sig.istft_filtered(z, h, w, opts) == Sig.istft(Nx.multiply(z, h), w, opts)
  end

  test "the dispatch record names the kernel family of the last call (round 6)" do
    ctx = Sig.context()
    w = Sig.Windows.hann(1024)
    x = Nx.iota({48_000}, type: :f32) |> Nx.sin()
    {z, _t, _f} = Sig.stft(x, w, overlap_length: 768, fft_length: 1024, sampling_rate: 48_000)
    assert String.starts_with?(Sig.last_dispatch(ctx), "stft.pair")
    _ = Sig.istft(z, w, overlap_length: 768, fft_length: 1024, sampling_rate: 48_000)
    assert String.starts_with?(Sig.last_dispatch(ctx), "istft.wave")
    h = Sig.Filters.firwin(4097, [0.2])
    _ = Sig.Filters.fir(Nx.iota({100_000}, type: :f32) |> Nx.cos(), h, mode: :same)
    assert String.starts_with?(Sig.last_dispatch(ctx), "fir.dline")
  end

  test "the sharded log-mel all-reduces the global maximum: channel shards equal the unsharded call" do
    g = Sig.Sharded.group()
    x = Nx.iota({4, 30_000}, type: :f32) |> Nx.sin() |> Nx.multiply(Nx.tensor([[1.0e-3], [1.0e-3], [1.0], [1.0]]))
    w = Sig.Windows.hann(1024)
    opts = [overlap_length: 768, sampling_rate: 48_000, mel_bins: 80]
    assert Sig.Sharded.mel_spectrogram(g, x, w, opts) == Sig.mel_spectrogram(x, w, opts)
  end
end
